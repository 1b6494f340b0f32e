"""Keep the process that drives a GPU on the CPU cores of that GPU's NUMA node.

The fitting step is ~170 kernel launches issued from Python; at BASELINE configs[3] sizes the GPU needs 4.4 ms for them and the
interpreter 5-6 ms, so the step time IS the host's time.  On the two-socket hosts of this pool a process the scheduler left on the
far socket ran the same step 10-15 % slower (5.7-6.1 vs 5.0-5.3 ms, tools/ab/host_regions.py under taskset) — every doorbell
write, pinned-memory read-back and allocation crosses the socket link.  One rank per GPU binds itself to its GPU's node (what
``numactl --cpunodebind`` does from outside; here without a launcher hop, which rocprofv3 forbids on this pool).
"""
from __future__ import annotations

import os

_BOUND = {}
_PREVIOUS = {}      # device index -> the affinity mask bind_to_device replaced (what unbind() puts back)


def _cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_cpus(device_index: int):
    """CPUs local to the GPU's PCI device (sysfs ``local_cpulist``), or None when that cannot be read."""
    import torch
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            cpus = _cpulist(f.read())
        return cpus or None
    except Exception:
        return None


def bind_to_device(device=None) -> bool:
    """Restrict the calling process (this thread and the threads it creates from now on — the autograd engine's among them) to the
    CPUs of the GPU's NUMA node.  Never widens an affinity mask that is already narrower (a launcher's own binding wins); does
    nothing under GSVC_NO_CPU_BIND=1, off Linux, or when sysfs does not say.  Returns True when a mask was set.

    The mask stays until ``unbind()`` (``Trainer.close()`` / leaving ``with Trainer(...)``) puts the previous one back: anything
    else the process runs meanwhile — entropy coding, OpenMP code, DataLoader workers created later — is limited to that node's
    cores, and threads that already existed when this was called (an autograd engine thread from an earlier backward) keep
    their own mask."""
    import torch
    if os.environ.get("GSVC_NO_CPU_BIND") or not hasattr(os, "sched_setaffinity") or not torch.cuda.is_available():
        return False
    idx = torch.cuda.current_device() if device is None else (device if isinstance(device, int) else (torch.device(device).index or 0))
    if idx in _BOUND:
        return _BOUND[idx]
    ok = False
    local = device_cpus(idx)
    if local:
        try:
            have = os.sched_getaffinity(0)
            want = have & local
            if want and want != have:
                os.sched_setaffinity(0, want)
                _PREVIOUS[idx] = have
                ok = True
        except OSError:
            ok = False
    _BOUND[idx] = ok
    return ok


def unbind(device=None) -> bool:
    """Put back the affinity mask ``bind_to_device`` replaced (calling thread; threads created while bound keep the narrow mask)."""
    import torch
    if not _PREVIOUS:
        return False
    idx = next(iter(_PREVIOUS)) if device is None else (device if isinstance(device, int) else (torch.device(device).index or 0))
    have = _PREVIOUS.pop(idx, None)
    _BOUND.pop(idx, None)
    if have is None:
        return False
    try:
        os.sched_setaffinity(0, have)
        return True
    except OSError:
        return False
