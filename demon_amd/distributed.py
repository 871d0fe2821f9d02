"""Multi-GPU plumbing: one process per GPU.  The path shards by image pair with no data-path collective
(SURVEY.md section 8e); the only collective is one broadcast of the weights at start-up.

Two broadcast routes:
  * "rccl"  (GPU, default): an RCCL communicator created THROUGH THE C ABI (demon_comm_*: ncclGetUniqueId /
            ncclCommInitRank) and ONE ncclBroadcast of the packed, device-resident weight slab (demon_broadcast_weights).
            torch.distributed is used only to ship the 128-byte unique id between the ranks (any other channel works:
            `exchange` is a callable), so a ctypes / C caller can bring up N GPUs without torch.
  * "torch" (CPU tests with gloo, or two ranks sharing one GPU, which RCCL refuses): torch.distributed.broadcast of the
            TF-layout float blob, then demon_set_weights_blob(_device) on every rank.
"""
import ctypes

import numpy as np


# ---- host placement: a rank's host threads and pinned buffers next to ITS GPU ------------------------------------------------------------
# One process per GPU on an 8-GPU node: every rank feeds ~240-launch graphs (4 lanes) and, host to host, copies 39 MB in / 10 MB out per
# step through page-locked buffers.  The GPU hangs off one NUMA node; a rank whose thread (and whose pinned pages, first touched by that
# thread) lives on the other socket pays a cross-socket hop on every doorbell and every DMA (SURVEY.md section 8e "watch").  The node
# comes from sysfs -- /sys/class/drm/card*/device/numa_node of the PCI device the HIP runtime names for the ordinal -- and the binding is
# plain sched_setaffinity on that node's cpulist (memory then follows first touch; no libnuma in the image).  Everything degrades to a
# no-op with a reason: no sysfs entry, node -1 (single-socket or virtualised box), an affinity mask that excludes the node.
def gpu_pci_bus_id(device):
    """'0000:c1:00.0' of HIP device `device` (torch if it is loaded, else rocm-smi-free sysfs order), or None"""
    import sys
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_available():
            p = torch.cuda.get_device_properties(device)
            if hasattr(p, "pci_bus_id"):
                return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:
        pass
    return None


def gpu_numa_node(device, sysfs="/sys"):
    """(numa node, how it was found) of HIP device `device`; node None when unknown"""
    import glob
    import os
    bus = gpu_pci_bus_id(device)
    if bus:
        path = os.path.join(sysfs, "bus/pci/devices", bus, "numa_node")
        try:
            with open(path) as f:
                return int(f.read().strip()), path
        except (OSError, ValueError):
            pass
    # fallback: the device-th AMD render node in sysfs order (vendor 0x1002), which is the order the runtime enumerates on one node
    cards = []
    for d in sorted(glob.glob(os.path.join(sysfs, "class/drm/renderD*/device"))):
        try:
            with open(os.path.join(d, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
            with open(os.path.join(d, "numa_node")) as f:
                cards.append((int(f.read().strip()), os.path.join(d, "numa_node")))
        except (OSError, ValueError):
            continue
    if 0 <= device < len(cards):
        return cards[device]
    return None, "no sysfs entry for device %d" % device


def parse_cpulist(text):
    """'0-63,128-191' -> sorted list of ints"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def bind_to_gpu_numa_node(device, sysfs="/sys", apply=True):
    """Restricts this process to the CPUs of the NUMA node HIP device `device` is attached to (sched_setaffinity; memory follows
    first touch, so call it BEFORE allocating / pinning host buffers).  Returns a record for the bench line:
    {"numa_node", "cpus": count, "bound": bool, "why"}.  Never raises."""
    import os
    rec = {"numa_node": None, "cpus": None, "bound": False, "why": ""}
    try:
        node, how = gpu_numa_node(device, sysfs)
        rec["numa_node"] = node
        if node is None or node < 0:
            rec["why"] = "no NUMA node for the GPU (%s)" % how
            return rec
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as f:
            cpus = parse_cpulist(f.read())
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        rec["cpus"] = len(allowed)
        if not allowed:
            rec["why"] = "the process's affinity mask has no CPU of node %d" % node
            return rec
        if apply:
            os.sched_setaffinity(0, allowed)
        rec["bound"] = bool(apply)
        rec["why"] = "node %d from %s" % (node, how)
    except Exception as e:
        rec["why"] = "binding failed: %r" % (e,)
    return rec


def shard_range(global_batch, rank, world):
    """rank r of `world` owns pairs [lo, hi); remainder pairs go to the lowest ranks."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def torch_exchange(payload, src=0):
    """ships a small bytes object from rank `src` to every rank over the initialised torch.distributed group"""
    import torch.distributed as dist
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


class NativeComm:
    """RCCL communicator owned through the C ABI (include/demon_hip.h: demon_comm_*).  `exchange(bytes_or_None) -> bytes`
    ships rank 0's unique id to everybody (default: torch.distributed).

    The id exchange is UNCONDITIONAL: rank 0 always ships something -- the 128-byte id, or a marker `b"!" + reason` when it could
    not create one (librccl not loadable, ncclGetUniqueId failed) -- so the other ranks, who are blocked in `exchange`, never wait
    for a rank 0 that has already given up; every rank then raises the same error and can fall back together."""

    def __init__(self, rank, world, device, exchange=torch_exchange):
        from . import _lib
        self.lib = _lib.load()
        self.rank, self.world, self.device = rank, world, device
        self.handle = None
        payload = None
        if rank == 0:
            try:   # whatever goes wrong here, rank 0 still ships a marker: the other ranks are about to block in `exchange`
                buf = ctypes.create_string_buffer(128)
                rc = self.lib.demon_comm_get_unique_id(buf)
                payload = buf.raw if rc == 0 else b"!demon_comm_get_unique_id failed (%d): %s" % (rc, self.lib.demon_last_error(None) or b"")
            except Exception as e:
                rc, payload = -1, b"!" + repr(e).encode(errors="replace")
            if rc != 0 and len(payload) == 128:
                payload += b" "   # an id is exactly 128 bytes, the marker never is
        uid = exchange(payload) if world > 1 else payload
        if uid[:1] == b"!" and len(uid) != 128:
            raise RuntimeError("rank 0 could not create an RCCL id: " + uid[1:].decode(errors="replace"))
        handle = ctypes.c_void_p()
        self._check(self.lib.demon_comm_init_rank(ctypes.byref(handle), world, uid, rank, device), "demon_comm_init_rank")
        self.handle = handle

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.demon_last_error(None).decode()))

    def count(self):
        """ncclCommCount: the number of ranks RCCL itself reports for this communicator"""
        n = ctypes.c_int(0)
        self._check(self.lib.demon_comm_count(self.handle, ctypes.byref(n)), "demon_comm_count")
        return int(n.value)

    def broadcast_weights(self, ctx, root=0):
        """one ncclBroadcast of ctx's packed weight slab from `root`; every rank must call it"""
        rc = self.lib.demon_broadcast_weights(ctx.h, self.handle, root, self.rank)
        if rc != 0:
            raise RuntimeError("demon_broadcast_weights failed (%d): %s" % (rc, self.lib.demon_last_error(ctx.h).decode()))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.demon_comm_destroy(self.handle)
            self.handle = None


def all_ranks_ok(ok, device_index=None):
    """MIN over the ranks of a 0 / 1 flag through the initialised torch.distributed group; the flag tensor lives where the
    group's backend can reduce it (CPU for gloo, this rank's GPU for nccl = RCCL)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    backend = str(dist.get_backend()).lower()
    dev = torch.device("cuda", device_index if device_index is not None else torch.cuda.current_device()) if "nccl" in backend else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def broadcast_blob(blob, nfloats, device, src=0):
    """blob: 1-D float32 numpy array on `src`, ignored elsewhere.  Returns a torch tensor on `device`
    holding the blob on every rank (torch.distributed.broadcast: gloo on CPU tensors, RCCL on GPU tensors)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device)
            if t.numel() != nfloats:
                raise ValueError("blob has %d floats, expected %d" % (t.numel(), nfloats))
        else:
            t = torch.empty(nfloats, dtype=torch.float32, device=device)
        dist.broadcast(t, src=src)
        return t
    return torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device)


LAST_BROADCAST = {}   # what the most recent distribute_weights call did on this rank: route, rccl_nranks, seconds (bench.py prints it)


def distribute_weights(ctx, host_weights, rank, world, route="rccl", comm=None):
    """Puts rank 0's weights (dict tf name -> array; None on the other ranks) on every rank's context.
    Returns (seconds spent in the broadcast, route description)."""
    import time
    from . import weights as W
    LAST_BROADCAST.clear()
    LAST_BROADCAST.update(route=route, rccl_nranks=None, seconds=0.0)
    if route == "rccl":
        own = comm is None
        note = ""
        if own:
            # every rank must end up on the same route.  NativeComm's id exchange is unconditional (rank 0 ships an error marker
            # instead of an id when it has none), so all ranks reach the agreement below having run the same collectives
            nranks = None
            try:
                comm = NativeComm(rank, world, ctx.device)
                nranks = comm.count()   # (inside the try: a failure on ONE rank must reach the consensus below, not raise past it)
                ok = 1
            except Exception as e:   # librccl missing, rank 0 without an id, ncclCommInitRank or ncclCommCount failed
                if comm is not None:
                    comm.close()
                comm, ok, note = None, 0, " (native RCCL init failed on rank %d: %s)" % (rank, e)
            if not all_ranks_ok(ok, ctx.device):
                if comm is not None:
                    comm.close()
                dt, desc = distribute_weights(ctx, host_weights, rank, world, route="torch-gpu")
                return dt, desc + " [fallback]" + note
        else:
            nranks = comm.count()
        if rank == 0:
            ctx.set_weights(host_weights)      # TF layouts -> packed slab, on the root only
        t0 = time.perf_counter()
        comm.broadcast_weights(ctx, 0)         # synchronous on the context's stream
        dt = time.perf_counter() - t0
        if own:
            comm.close()
        LAST_BROADCAST.update(rccl_nranks=nranks, seconds=dt)
        return dt, "one ncclBroadcast of the packed %.0f MB weight slab through the C ABI (demon_broadcast_weights, %d RCCL ranks)" % (
            ctx.lib.demon_weights_slab_bytes(ctx.h) / 1e6, nranks)
    import torch
    order = ctx.variables()
    blob = W.weights_to_blob(host_weights, order) if rank == 0 else None
    on_gpu = route == "torch-gpu"
    t0 = time.perf_counter()
    t = broadcast_blob(blob, ctx.blob_size(), torch.device("cuda", ctx.device) if on_gpu else "cpu")
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if on_gpu:
        ctx.set_weights_blob_device(t.data_ptr(), t.numel())
    else:
        ctx.set_weights_blob(t.numpy())
    LAST_BROADCAST.update(route=route, seconds=dt)
    return dt, "torch.distributed.broadcast of the %.0f MB TF-layout blob (%s)" % (4e-6 * ctx.blob_size(), route)


def max_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(value)
