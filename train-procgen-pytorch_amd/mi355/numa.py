"""Keep a rank's host threads and its pinned buffers on the NUMA node its GPU hangs off.

The rollout is a chain of host <-> device hand-offs 257 times per iteration (ticket in host memory, doorbells, a 0.8-3 MB upload per env
group): with the process free to run on either socket the scheduler puts the spinning threads -- and, by first touch, the page-locked
buffers -- on the far node about half the time.  Called once, before the engine allocates anything; does nothing when the topology
cannot be read or the node offers fewer than `min_cpus` of the CPUs this process may use."""
import os


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(device_index=0):
    """CPUs of the NUMA node of GPU `device_index` (torch's numbering), or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return _parse_cpulist(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read()) or None
    except Exception:
        return None


def pin_to_gpu_node(device_index=0, min_cpus=8):
    """Restrict this process (and the threads it starts later) to the GPU's node.  Returns the CPU set used, or None."""
    if os.environ.get("MI355_NUMA_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    local = gpu_numa_cpus(device_index)
    if not local:
        return None
    want = local & os.sched_getaffinity(0)
    if len(want) < min_cpus:
        return None
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return None
    # sched_setaffinity(0, ...) moves the CALLING thread only; threads that already exist (torch's intra-op pool, a process group's
    # watchdog started by init_process_group) keep their masks: apply it to every thread of the process
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), want)
            except OSError:
                pass
    except OSError:
        pass
    return want
