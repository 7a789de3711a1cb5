"""Host-thread placement: the threads of one rank on a compact block of cores next to its GPU.

The host side of the path (tray building, family selection, NAST finish, result copies) is a pool of
about ten threads that pass tens of MB per batch between each other and to pinned staging buffers.
Left to the scheduler on a 2-socket / 16-CCD host they wander over 256 logical CPUs; kept on
sixteen neighbouring physical cores of the GPU's NUMA node (two CCDs: shared L3, local memory) the
same work costs less CPU and the host-bound workloads run faster (measured on the MI355X box,
2 x EPYC 9575F: V4 amplicons 244 k -> 289 k sequences/s, 16S 110 k -> 112 k at 9.2 -> 7.4 busy cores).

pin_rank() is called once per process, before the host library creates its threads (they inherit
the calling thread's mask).  Ranks that share a NUMA node split its cores into equal blocks.
SINA_HOST_PIN=0 disables it; any surprise in sysfs leaves the affinity untouched.
"""
import os

CORES_PER_RANK = 16


def _parse_cpulist(text):
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


def _physical_cores(cpus):
    """One logical CPU per physical core (the lowest-numbered sibling), ascending."""
    firsts = set()
    for c in cpus:
        try:
            sib = _parse_cpulist(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read())
            firsts.add(min(sib))
        except OSError:
            firsts.add(c)
    return sorted(firsts)


def gpu_numa_node(props):
    """NUMA node of a device from its torch device properties (PCI address -> sysfs), or None."""
    try:
        bus = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        return node if node >= 0 else None
    except (OSError, ValueError, AttributeError):
        return None


def plan(local_rank, nodes, node_cpus, allowed, cores_per_rank=CORES_PER_RANK):
    """CPUs for `local_rank`.  nodes[j] = NUMA node of local device j (None = unknown);
    node_cpus[n] = logical CPUs of node n as one-per-physical-core list; allowed = current mask.
    Returns a sorted list, or None to leave the affinity alone."""
    node = nodes[local_rank]
    if node is None or node not in node_cpus:
        # no NUMA information for this device (containers often hide it): compactness is most of the
        # gain -- the ranks without a node split all allowed cores, in order
        if None not in node_cpus:
            return None
        node = None
    cores = [c for c in node_cpus[node] if c in allowed]
    sharers = [j for j, n in enumerate(nodes) if (n if n in node_cpus else None) == node]
    k, n = sharers.index(local_rank), len(sharers)
    per = len(cores) // n
    if per < 4:  # not enough cores on the node to be worth confining anything
        return None
    block = cores[k * per:(k + 1) * per][:cores_per_rank]
    return block or None


def pin_rank(local_rank, local_world):
    """Pins the calling thread (and every thread it creates from now on).  Returns the CPU list, or None."""
    if os.environ.get("SINA_HOST_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import torch
        nodes = [gpu_numa_node(torch.cuda.get_device_properties(j)) for j in range(local_world)]
        allowed = os.sched_getaffinity(0)
        node_cpus = {}
        for n in set(x for x in nodes if x is not None):
            try:
                node_cpus[n] = _physical_cores(_parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % n).read()))
            except OSError:
                pass
        if any(x not in node_cpus for x in nodes):  # one unknown: nobody uses NUMA nodes (blocks must not overlap)
            nodes = [None] * len(nodes)
            node_cpus = {None: _physical_cores(sorted(allowed))}
        cpus = plan(local_rank, nodes, node_cpus, allowed)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:  # noqa: BLE001 -- placement is an optimisation, never a reason to fail
        return None
