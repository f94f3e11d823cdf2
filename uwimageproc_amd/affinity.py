"""Host placement of a rank: which CPUs sit next to GPU `index`, read from sysfs WITHOUT touching the GPU (so it can run
in a fresh rank process before the first HIP call).  The 1080p pipe moves 1.6 GB per step each way through page-locked
host memory ("PCIe H2D/D2H of frames is the inter-device bottleneck to watch"): a rank whose threads
-- and therefore its first-touched pinned buffers -- live on the GPU's NUMA node keeps that traffic off the inter-socket
link.  Everything here degrades to "unavailable" when the container hides sysfs."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
PCI_DEVICES = "/sys/bus/pci/devices"


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11]"""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _props(path: str) -> Dict[str, int]:
    d: Dict[str, int] = {}
    with open(path) as f:
        for line in f:
            kv = line.split()
            if len(kv) == 2:
                try:
                    d[kv[0]] = int(kv[1])
                except ValueError:
                    pass
    return d


def gpu_pci_addresses(kfd_nodes: str = KFD_NODES) -> List[str]:
    """PCI addresses (dddd:bb:dd.f) of the KFD GPU nodes in topology order -- the order the HIP runtime enumerates
    devices in when no *_VISIBLE_DEVICES variable reorders them."""
    out = []
    for n in sorted((x for x in os.listdir(kfd_nodes) if x.isdigit()), key=int):
        try:
            p = _props(os.path.join(kfd_nodes, n, "properties"))
        except OSError:
            continue
        if p.get("simd_count", 0) <= 0:          # a CPU node
            continue
        loc, dom = p.get("location_id", 0), p.get("domain", 0)
        out.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    return out


def visible_index(index: int, env=os.environ) -> Optional[int]:
    """Map a HIP device index through ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES (plain integer lists only)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = env.get(var)
        if v:
            try:
                ids = [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None                       # UUID form: give up rather than guess
            if index >= len(ids):
                return None
            index = ids[index]
    return index


def gpu_local_cpus(index: int, kfd_nodes: str = KFD_NODES, pci_devices: str = PCI_DEVICES, env=os.environ) -> Dict:
    """{"numa_node": n, "cpus": [...], "pci": "dddd:bb:dd.f"} for HIP device `index`, or {"error": why}."""
    try:
        idx = visible_index(index, env)
        if idx is None:
            return {"error": "device order not derivable from *_VISIBLE_DEVICES"}
        gpus = gpu_pci_addresses(kfd_nodes)
        if idx >= len(gpus):
            return {"error": f"KFD topology lists {len(gpus)} GPUs, device {idx} asked"}
        bdf = gpus[idx]
        base = os.path.join(pci_devices, bdf)
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read().strip())
        with open(os.path.join(base, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        if not cpus:
            return {"error": "empty local_cpulist", "pci": bdf, "numa_node": node}
        return {"numa_node": node, "cpus": cpus, "pci": bdf}
    except OSError as e:
        return {"error": f"sysfs unavailable ({e.__class__.__name__}: {e.filename})"}


def pin_rank_to_gpu(index: int, ranks_on_node: int = 1, local_rank: int = 0, **kw) -> Dict:
    """Restrict the calling process (call it before any thread exists and before the first GPU call) to the CPUs local to
    its GPU, intersected with what the process may already use.  When several ranks' GPUs hang off the same NUMA node the
    node's CPUs are split evenly between those ranks.  Returns what was done, for the bench line."""
    info = gpu_local_cpus(index, **kw)
    allowed = sorted(os.sched_getaffinity(0))
    info["allowed_before"] = len(allowed)
    if "cpus" not in info:
        info["pinned"] = False
        return info
    mine = [c for c in info["cpus"] if c in set(allowed)]
    # ranks whose GPU shares this NUMA node: split its CPUs
    same = []
    for r in range(ranks_on_node):
        o = gpu_local_cpus(r, **kw)
        if o.get("numa_node") == info["numa_node"] and "cpus" in o:
            same.append(r)
    if len(same) > 1 and local_rank in same and len(mine) >= len(same):
        per = len(mine) // len(same)
        j = same.index(local_rank)
        mine = mine[j * per:(j + 1) * per]
    if not mine:
        info["pinned"] = False
        info["error"] = "the GPU's local CPUs are outside this process's affinity mask"
        return info
    os.sched_setaffinity(0, mine)
    info["pinned"] = True
    info["cpus"] = len(mine)
    info["cpu_range"] = f"{mine[0]}-{mine[-1]}"
    return info
