"""Whole-job rate vs coder threads, with the cgroup's CPU usage and throttling counters around each run (host-bound job
under a CPU-time quota: does the quota bite?)."""
import json, os, subprocess, sys
def stat():
    d = {}
    for line in open("/sys/fs/cgroup/cpu.stat"):
        k, v = line.split(); d[k] = int(v)
    return d
for th in (12, 14, 15, 16, 18, 16, 14):
    a = stat()
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "32"], capture_output=True, text=True,
                       env=dict(os.environ, BSCGPU_HOST_THREADS=str(th)))
    b = stat()
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"threads={th:2d}  {d['value']:7.1f} MB/s  {d['ms_per_step']:6.2f} ms/step   whole process: cpu {1e-6*(b['usage_usec']-a['usage_usec']):6.1f} s, "
          f"periods {b['nr_periods']-a['nr_periods']}, throttled periods {b['nr_throttled']-a['nr_throttled']}, throttled {1e-6*(b['throttled_usec']-a['throttled_usec']):.2f} s")
