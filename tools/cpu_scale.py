import sys, time, os, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import c_oracle, soundml_oracle as O
c = O.stft_config(2048, hop=512)
x = np.random.default_rng(0).uniform(-1, 1, (256, 480000)).astype(np.float32)
print("cpus", os.cpu_count())
os.system("lscpu | grep -i 'model name\\|socket\\|numa node(s)\\|thread(s) per core\\|MHz' | head -8")
t0 = time.perf_counter(); c_oracle.stft(c, x[:1], 2.0, threads=1); print("1 clip 1 thread %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); c_oracle.stft(c, x[:1], 2.0, threads=1); print("1 clip 1 thread %.3f s (again)" % (time.perf_counter() - t0))
for th in (16, 32, 64, 128, 256, 128, 64):
    t0 = time.perf_counter(); c_oracle.stft(c, x, 2.0, threads=th); dt = time.perf_counter() - t0
    print("threads %3d: %.3f s  %.2f Mframes/s" % (th, dt, 256 * 938 / dt / 1e6))
print("affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
import resource
r0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.perf_counter(); c_oracle.stft(c, x, 2.0, threads=64); dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF)
print("64 threads: wall %.3f s, user %.3f s, sys %.3f s, minor faults %d, invol ctx %d" % (dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_minflt - r0.ru_minflt, r1.ru_nivcsw - r0.ru_nivcsw))
out = np.empty((256, 1025, 938), np.float32); out.fill(0)
r0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.perf_counter(); c_oracle.stft(c, x, 2.0, threads=64, out=out); dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF)
print("64 threads, mapped out: wall %.3f s, user %.3f s, sys %.3f s, minor faults %d, invol ctx %d" % (dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_minflt - r0.ru_minflt, r1.ru_nivcsw - r0.ru_nivcsw))
