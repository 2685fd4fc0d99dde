"""Where is the GPU, where may this process run?  python scratch/numa_probe.py"""
import os, glob, torch
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
print("gpu", p.name, "bdf", bdf)
for f in ("numa_node", "local_cpulist"):
    try:
        print(f, open(f"/sys/bus/pci/devices/{bdf}/{f}").read().strip())
    except Exception as e:
        print(f, "unreadable:", e)
aff = sorted(os.sched_getaffinity(0))
print("affinity", len(aff), "cpus:", aff[:8], "...", aff[-4:])
for n in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    print(os.path.basename(n), open(n + "/cpulist").read().strip())
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
