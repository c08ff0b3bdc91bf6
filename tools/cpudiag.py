import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads(), flush=True)
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup", e)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)' ; free -g | head -2")
for nt in (torch.get_num_threads(), 16, 32, 64):
    torch.set_num_threads(nt)
    x = torch.empty(100_000_000)
    t0 = time.time(); x.normal_(); t1 = time.time()
    a = torch.randn(4096, 4096); b = torch.randn(4096, 4096)
    t2 = time.time(); (a @ b); t3 = time.time()
    c = torch.randn(8, 320, 64, 64); w = torch.randn(320, 320, 3, 3)
    t4 = time.time(); torch.nn.functional.conv2d(c, w, padding=1); t5 = time.time()
    print(f"threads {nt}: normal_ 100M {t1-t0:.2f}s  matmul4096 {t3-t2:.2f}s ({2*4096**3/(t3-t2)/1e12:.2f} TF)  conv {t5-t4:.2f}s", flush=True)
