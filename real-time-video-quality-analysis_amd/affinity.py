"""CPU affinity of a process that feeds one GPU: the CPUs of the GPU's NUMA node, from sysfs alone.

With eight ranks feeding eight GPUs over PCIe (BASELINE.json configs[4]: one 2160p stream per GPU), a copier thread of
the pinned ring (stream.py) that runs on the far socket moves its bytes across the inter-socket link first.  bind_numa()
pins the calling process - and every thread it starts afterwards - to the GPU's local CPUs.  It must run BEFORE the
first GPU call of the process and makes none itself (no HIP, no torch): bench.py calls it right after its CPU baseline.
The reference has no counterpart (its workers are a process pool on whatever cores the OS picks, complexity_metrics.py:143).
"""
import os


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _visible_index(index):
    """HIP device `index` of this process -> the node's device number (ROCR_/HIP_/CUDA_VISIBLE_DEVICES as plain index lists)"""
    vis = next((os.environ[k] for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(k)), None)
    if vis is None:
        return index, None
    try:
        return [int(x) for x in vis.split(",")][index], None
    except (ValueError, IndexError):
        return None, "device visibility list %r is not a plain index list" % vis


def _cpus_from_kfd(index, sysfs):
    """KFD topology nodes in node order are the HIP devices in device order; each names its PCI function (domain,
    location_id = bus << 8 | devfn), whose local_cpulist is the answer."""
    nodes_dir = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    gpus = []
    for node in sorted(os.listdir(nodes_dir), key=int):
        props = dict(line.split()[:2] for line in open(os.path.join(nodes_dir, node, "properties")) if len(line.split()) >= 2)
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(props)
    pr = gpus[index]
    loc, dom = int(pr["location_id"]), int(pr.get("domain", "0"))
    path = os.path.join(sysfs, "bus", "pci", "devices", "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7),
                        "local_cpulist")
    return _parse_cpulist(open(path).read()), path


def _cpus_from_drm(index, sysfs, dev_dri):
    """Where the KFD topology is not readable (an unprivileged container: EPERM on .../properties, seen on the GPU boxes
    of this pool): the render nodes the process can OPEN (/dev/dri/renderD*: a container is given exactly its GPUs'), in
    minor order, that are PCI functions (partition nodes - amdgpu_xcp platform devices - have no local_cpulist)."""
    minors = sorted(int(n[7:]) for n in os.listdir(dev_dri) if n.startswith("renderD") and n[7:].isdigit()
                    and os.access(os.path.join(dev_dri, n), os.R_OK | os.W_OK))
    paths = [os.path.join(sysfs, "class", "drm", "renderD%d" % m, "device", "local_cpulist") for m in minors]
    paths = [p for p in paths if os.path.isfile(p)]
    path = paths[index]
    return _parse_cpulist(open(path).read()), path


def gpu_local_cpus(index, sysfs="/sys", dev_dri="/dev/dri"):
    """The CPUs local to HIP device `index` (its NUMA node), from sysfs alone - no GPU call.
    -> (set of cpus, source path) or (None, reason)."""
    index, why = _visible_index(index)
    if index is None:
        return None, why
    errs = []
    for name, fn in (("kfd", lambda: _cpus_from_kfd(index, sysfs)), ("drm", lambda: _cpus_from_drm(index, sysfs, dev_dri))):
        try:
            cpus, path = fn()
            if cpus:
                return cpus, path
            errs.append("%s: empty %s" % (name, path))
        except (OSError, ValueError, KeyError, IndexError) as e:
            errs.append("%s: %s" % (name, e))
    return None, "; ".join(errs)


def bind_numa(device, enabled=True, sysfs="/sys", dev_dri="/dev/dri"):
    """--bind-numa: pin this rank (and the copier threads it will start) to the CPUs of its GPU's NUMA node BEFORE any GPU
    call.  With eight ranks feeding eight GPUs over PCIe, a copier thread on the far socket halves its bandwidth.
    -> the record for config.cpu_affinity."""
    if not enabled:
        return {"bound": False, "why": "--no-bind-numa"}
    if not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "why": "no sched_setaffinity on this platform"}
    cpus, src = gpu_local_cpus(device, sysfs, dev_dri)
    if cpus is None:
        return {"bound": False, "why": src}
    have = os.sched_getaffinity(0)
    want = cpus & have
    if not want:
        return {"bound": False, "why": "the GPU's local CPUs %s are outside this process's affinity" % sorted(cpus)[:4]}
    if want != have:
        os.sched_setaffinity(0, want)
    return {"bound": True, "cpus": len(want), "of_visible": len(have), "first": min(want), "last": max(want), "source": src}
