#!/bin/bash
# first-contact probe of the GPU box: what host/GPU do we actually have?
mkdir -p gpurun_out
{
  echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" ; free -g | head -2
  echo "== rocm-smi"; rocm-smi --showproductname --showmeminfo vram 2>&1 | head -30
  echo "== rocminfo gfx"; rocminfo 2>/dev/null | grep -E "gfx|Compute Unit|Marketing" | head -12
  python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available(), torch.cuda.device_count(), torch.cuda.get_device_name(0))"
} > gpurun_out/probe.txt 2>&1
cat gpurun_out/probe.txt
