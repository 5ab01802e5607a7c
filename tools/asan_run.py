"""Run a command on the sanitizer build of the library (python -m libbsc_amd.build --asan): AddressSanitizer + UBSan on all
host code, kernels unchanged.  The command's python / CLI stays uninstrumented; the runtime is preloaded.
    python tools/asan_run.py python -m pytest tests -q -m "not gpu"
    python tools/asan_run.py python tools/fuzz_gpu.py 300 303
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libbsc_amd.build import asan_env, build, ASAN_LIB

if not os.path.exists(ASAN_LIB):
    build(asan=True)
env = asan_env()
os.execvpe(sys.argv[1], sys.argv[1:], env)
