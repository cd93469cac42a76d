"""CPU affinity per GPU rank for AMD hosts: the API of the reference's utils/gpu_affinity.py (set_affinity and its five
modes, :126-146), with the NVML topology query replaced by Linux sysfs: a GPU's NUMA-local cores are
/sys/bus/pci/devices/<domain:bus:dev.fn>/local_cpulist of its PCI function (bus id from the HIP device properties)."""
import collections
import os
import pathlib
import re


def parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def device_pci_bdf(gpu_id):
    import torch
    p = torch.cuda.get_device_properties(gpu_id)
    return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)


def device_cpu_affinity(gpu_id, sysfs="/sys/bus/pci/devices"):
    """NUMA-local logical CPUs of the GPU (all online CPUs if the topology is not exposed)"""
    try:
        with open(os.path.join(sysfs, device_pci_bdf(gpu_id), "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        if cpus:
            return cpus
    except Exception:
        pass
    return sorted(os.sched_getaffinity(0))


def get_thread_siblings_list(root="/sys/devices/system/cpu"):
    pairs = []
    pat = re.compile(r"(\d+)\D(\d+)")
    for fname in pathlib.Path(root).glob("cpu*/topology/thread_siblings_list"):
        m = pat.findall(fname.read_text().strip())
        if m:
            pairs.append(tuple(map(int, m[0])))
    return pairs


def plan_affinity(gpu_id, world_size, mode, affinity_of=device_cpu_affinity, siblings=None):
    """the CPU set `set_affinity` would apply (pure function: testable without GPUs)"""
    if mode == "socket":
        return list(affinity_of(gpu_id))
    if mode == "single":
        return list(affinity_of(gpu_id))[:1]
    sib = dict(get_thread_siblings_list() if siblings is None else siblings)
    socket_aff = [[c for c in affinity_of(i) if c not in set(sib.values())] for i in range(world_size)]
    if mode == "single_unique":
        taken, out = set(), []
        for aff in socket_aff:
            pick = next((c for c in aff if c not in taken), None)
            if pick is None:
                raise RuntimeError("not enough cores for one unique core per GPU")
            taken.add(pick)
            out.append([pick])
        return out[gpu_id]
    if mode in ("socket_unique_interleaved", "socket_unique_continuous"):
        groups = collections.defaultdict(list)
        for i, aff in enumerate(socket_aff):
            groups[tuple(aff)].append(i)
        for aff, devs in groups.items():
            if gpu_id not in devs:
                continue
            g, n = devs.index(gpu_id), len(devs)
            if mode.endswith("interleaved"):
                mine = list(aff[g::n])
            else:
                per = len(aff) // n
                mine = list(aff[g * per:(g + 1) * per])
            return mine + [sib[c] for c in mine if c in sib]
    raise RuntimeError("Unknown affinity mode")


def set_affinity(gpu_id=None, mode="socket"):
    if gpu_id is None:
        gpu_id = os.getenv("LOCAL_RANK", 0)
    gpu_id = int(gpu_id)
    world_size = int(os.getenv("LOCAL_WORLD_SIZE", os.getenv("WORLD_SIZE", 1)))
    cpus = plan_affinity(gpu_id, world_size, mode)
    if cpus:
        os.sched_setaffinity(0, cpus)
    return os.sched_getaffinity(0)
