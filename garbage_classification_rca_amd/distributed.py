"""Data parallelism for the MM-RCA training loop: one process per GPU, RCCL (torch.distributed backend "nccl" on
ROCm; "gloo" on CPU for tests).  Replaces the reference's single-process ``nn.DataParallel`` (main_both.py:386-388).

* samples shard by rank with a shared per-epoch permutation (``ShardedSampler``) -- no data-path collective;
* the one exchange step is the gradient all-reduce.  The gradient arena is flat and ordered
  text | vision | head, and the backward finishes spans in the order head, vision layers (last to first), text layers
  (last to first): ``GradSync`` issues one asynchronous all-reduce (average) per finished span on RCCL's own stream, so
  the exchange of layer i overlaps the backward of layers < i; ``finish()`` waits before the optimizer step.
  With gradient accumulation only the stepping micro-batch synchronises (main_both.py:116-124 semantics).
"""
from __future__ import annotations

import math
import os
import sys
from typing import Iterator, List, Optional

import torch
import torch.distributed as dist


def read_cpu_topology(cpus, sysfs: str = "/sys/devices/system") -> Optional[dict]:
    """{cpu: (numa node, (package, core id))} of the given CPUs from sysfs, or None when any of it cannot be read (then nothing is
    bound).  NUMA node = the node whose cpulist holds the CPU (a machine without NUMA has the one node0); package / core id from
    cpu*/topology.  Ids are NOT assumed contiguous: with SMT on, Linux numbers all physical cores first (socket 0, then socket 1) and
    their siblings after them -- 0-63 / 64-127 / 128-191 / 192-255 on a 2 x 64-core node are socket 0, socket 1, socket 0, socket 1."""
    def parse_list(text):
        out = []
        for part in text.strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
        return out
    try:
        node_of = {}
        node_dir = os.path.join(sysfs, "node")
        for name in sorted(os.listdir(node_dir)):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(node_dir, name, "cpulist")) as f:
                    for c in parse_list(f.read()):
                        node_of[c] = int(name[4:])
        topo = {}
        for c in cpus:
            base = os.path.join(sysfs, "cpu", f"cpu{c}", "topology")
            with open(os.path.join(base, "physical_package_id")) as f:
                pkg = int(f.read())
            with open(os.path.join(base, "core_id")) as f:
                core = int(f.read())
            topo[c] = (node_of[c], (pkg, core))
        return topo
    except (OSError, ValueError, KeyError):
        return None


def gpu_numa_nodes(n: int, sysfs_pci: str = "/sys/bus/pci/devices") -> Optional[List[int]]:
    """NUMA node of each of the first n GPUs (device order = LOCAL_RANK order) from /sys/bus/pci/devices/<bdf>/numa_node, or None when
    it cannot be told (no GPU, fewer than n devices, an unreadable file, a -1 entry: the firmware did not say)."""
    try:
        if not torch.cuda.is_available() or torch.cuda.device_count() < n:
            return None
        out = []
        for i in range(n):
            pr = torch.cuda.get_device_properties(i)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            with open(os.path.join(sysfs_pci, bdf, "numa_node")) as f:
                node = int(f.read())
            if node < 0:
                return None
            out.append(node)
        return out
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def cpu_slice(local_rank: int, local_world: int, cpus=None, topology=None, gpu_nodes=None) -> List[int]:
    """The CPU set of one rank on a node that runs `local_world` ranks, built from the machine's TOPOLOGY, not from the order of the
    CPU ids (contiguous eighths of the sorted ids put ranks 2-3 of a 2-socket SMT node on the wrong socket).
      * the allowed CPUs are grouped into physical cores (SMT siblings stay together) and the cores into NUMA nodes;
      * gpu_nodes (the NUMA node of every local rank's GPU, gpu_numa_nodes) known: the ranks whose GPU sits on a node split THAT node's
        cores among themselves, in rank order -- launch thread, DataLoader workers and pinned staging buffers next to their GPU;
      * gpu_nodes unknown: the cores, ordered by (node, package, core), are cut into local_world equal runs (ranks 0-3 on the first
        socket and 4-7 on the second of an 8-GPU node: how MI355X nodes are wired);
      * the topology unreadable: [] -- the caller then does not bind at all.
    With fewer cores than ranks the slices wrap (never empty when there is a CPU)."""
    cpus = sorted(os.sched_getaffinity(0) if cpus is None else cpus)
    local_world = max(int(local_world), 1)
    if not cpus:
        return []
    if topology is None:
        topology = read_cpu_topology(cpus)
    if topology is None or any(c not in topology for c in cpus):
        return []
    cores = {}
    for c in cpus:
        node, core = topology[c]
        cores.setdefault((node, core), []).append(c)
    ordered = sorted(cores)                       # (node, (package, core id))
    if gpu_nodes is not None and len(gpu_nodes) >= local_world:
        mine = gpu_nodes[local_rank]
        peers = [r for r in range(local_world) if gpu_nodes[r] == mine]
        pool = [k for k in ordered if k[0] == mine]
        idx, parts = peers.index(local_rank), len(peers)
        if not pool:
            return []
    else:
        pool, idx, parts = ordered, local_rank, local_world
    if len(pool) < parts:
        chosen = [pool[idx % len(pool)]]
    else:
        per = len(pool) // parts
        chosen = pool[idx * per: (idx + 1) * per]
    return sorted(c for k in chosen for c in cores[k])


MIN_CPUS_PER_RANK = 4


def bind_rank_to_cpus(local_rank: int, local_world: int) -> Optional[List[int]]:
    """Pin this process (and the DataLoader workers it will fork) to its rank's CPU slice; MMRCA_CPU_BIND=0 leaves the affinity
    alone.  Returns the slice, or None when nothing was changed (one rank per node or LOCAL_WORLD_SIZE unknown, binding off, the
    topology unreadable, fewer than MIN_CPUS_PER_RANK CPUs per rank, or no sched_setaffinity)."""
    if local_world <= 1 or os.environ.get("MMRCA_CPU_BIND", "1") != "1" or not hasattr(os, "sched_setaffinity"):
        return None
    sl = cpu_slice(local_rank, local_world, gpu_nodes=gpu_numa_nodes(local_world))
    if len(sl) < MIN_CPUS_PER_RANK:        # a rank also runs RCCL's proxy thread and the HIP runtime's helpers: never squeeze it onto < 4 cores
        return None
    try:
        os.sched_setaffinity(0, sl)
    except OSError:
        return None
    return sl


def loader_workers(requested: int) -> int:
    """DataLoader worker processes of this rank: the requested count (main_both.py:476-492 asks for 16), capped by the CPUs this
    process may run on minus one for the launch thread -- after bind_rank_to_cpus that is the rank's own slice, so eight ranks on a
    64-core node start 8 x 7 workers, not 8 x 16 that time-share."""
    if requested <= 0:
        return 0
    try:
        have = len(os.sched_getaffinity(0))
    except AttributeError:
        have = os.cpu_count() or 1
    return max(1, min(int(requested), have - 1))


def init_from_env(backend: Optional[str] = None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torchrun; returns (rank, local_rank, world).  With several ranks on the node
    (LOCAL_WORLD_SIZE) the process is pinned to its slice of the node's CPUs first (bind_rank_to_cpus)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (LOCAL_WORLD_SIZE unset -- a launcher that does not say how many ranks share this node: no binding, not a 1/world guess)
    bound = bind_rank_to_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", "0")))
    if bound is not None and rank == 0:
        print(f"[mmrca] rank -> CPU binding on: {len(bound)} CPUs per rank (rank 0: {bound[0]}..{bound[-1]}); MMRCA_CPU_BIND=0 disables",
              file=sys.stderr)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("MMRCA_DIST_BACKEND", backend)      # rehearsal override (e.g. gloo on a one-GPU box)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)         # several ranks may share a card in a rehearsal
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)
    return rank, local, world


class ShardedSampler(torch.utils.data.Sampler):
    """idx = perm(seed + epoch)[rank::world], padded (by wrapping) so that every rank draws the same count."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool = True, seed: int = 0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.per_rank = math.ceil(n / world) if n else 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def indices(self) -> List[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        total = self.per_rank * self.world
        if total > self.n and self.n:
            order = order + order[: total - self.n]
        return order[self.rank: total: self.world]

    def __iter__(self) -> Iterator[int]:
        return iter(self.indices())

    def __len__(self) -> int:
        return self.per_rank

    @property
    def num_real(self) -> int:
        """How many of this rank's ``per_rank`` draws are real samples: the wrap-around padding that equalises the ranks sits at
        the END of the padded order, i.e. in the LAST draws of the ranks it reaches.  Evaluation counts only the first
        ``num_real`` draws (otherwise up to world-1 samples are scored twice)."""
        return len(range(self.rank, self.n, self.world))


class BalancedShardedSampler(torch.utils.data.Sampler):
    """``--balanced_sampler`` (reference main_both.py:478-526 -> imbalanced_sampler/imbalanced.py:9-74): every draw picks sample i
    with probability proportional to 1 / count[label_i], WITH replacement, ``len(dataset)`` draws per epoch (so each class
    contributes the same expected number of draws).  Under data parallelism one multinomial draw of the whole epoch is made
    from the shared seed (+ epoch) and strided by rank, so the ranks together see exactly what one process would.  The
    reference uses it for the validation loaders too (every pass re-draws); ``num_real`` equals ``len`` -- there is no padding
    to exclude because the draws are random anyway."""

    def __init__(self, targets, rank: int, world: int, seed: int = 0):
        t = torch.as_tensor(list(targets), dtype=torch.int64)
        self.n, self.rank, self.world, self.seed, self.epoch = int(t.numel()), rank, world, seed, 0
        counts = torch.bincount(t) if self.n else torch.zeros(1, dtype=torch.int64)
        self.weights = (1.0 / counts[t].double()) if self.n else torch.zeros(0, dtype=torch.float64)     # imbalanced.py:42-46
        self.per_rank = math.ceil(self.n / world) if self.n else 0
        self._pass = 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def indices(self) -> List[int]:
        if not self.n:
            return []
        g = torch.Generator().manual_seed(self.seed + self.epoch + 7919 * self._pass)
        total = self.per_rank * self.world
        order = torch.multinomial(self.weights, total, replacement=True, generator=g).tolist()          # imbalanced.py:70-71
        return order[self.rank: total: self.world]

    def __iter__(self) -> Iterator[int]:
        self._pass += 1            # every pass over the loader re-draws, as the reference's global-RNG multinomial does
        return iter(self.indices())

    def __len__(self) -> int:
        return self.per_rank

    @property
    def num_real(self) -> int:
        return self.per_rank


class GradSync:
    """Overlapped gradient averaging over contiguous slices of a flat gradient buffer."""

    def __init__(self, flat_grads: torch.Tensor, world: Optional[int] = None, bucket_bytes: int = 32 << 20,
                 wire_dtype: Optional[torch.dtype] = None, force: bool = False):
        """wire_dtype=torch.bfloat16 (or MMRCA_GRAD_WIRE=bf16): each span is cast to bf16 on the producer stream, reduced in
        bf16 (half the bytes on xGMI: 304 MB instead of 609 MB per step for configs[1]) and written back into the fp32
        arena by finish().  Default: fp32 on the wire (bit-identical replicas, exact average).
        force: issue the collectives even when world == 1 (a one-GPU box can then drive the RCCL path itself -- ReduceOp.AVG, the
        bf16 wire, producer streams -- which is what tests/test_rccl_gpu.py does)."""
        self.g = flat_grads
        self.force = force
        if wire_dtype is None and os.environ.get("MMRCA_GRAD_WIRE", "fp32") == "bf16":
            wire_dtype = torch.bfloat16
        self.wire_dtype = wire_dtype
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // flat_grads.element_size())
        self.pending = []
        self.enabled = True
        self._acc_lo, self._acc_hi = None, None
        self.bytes_reduced = 0
        self.launches = 0
        # bench.py's self-diagnosis: with `timing` on, finish() brackets its waits with events on the compute stream
        self.timing, self.wait_events = False, []

    def active(self) -> bool:
        """does this exchange launch collectives at all (more than one rank, or forced on one)"""
        return self.world > 1 or self.force

    def capturable(self) -> bool:
        """can the exchange be recorded INSIDE a HIP graph of the train step (training.GraphedTrainStep)?  RCCL collectives can (the
        all-reduces run on RCCL's stream, forked from and joined back into the capturing stream by events: graph nodes like any
        other); gloo's cannot.  With MORE THAN ONE rank this is OPT-IN (MMRCA_GRAPH_DP=1): the captured exchange has only ever run on
        the world-1 RCCL group of a one-GPU box (tests/test_rccl_gpu.py), never on two ranks -- until it has, a multi-rank step is
        launched from Python.  (run_one_epoch makes the ranks agree on the caption width of every batch, agree_caption_width below, so
        that they key, warm up, capture and replay the same graphs at the same steps.)"""
        default = "1" if self.world == 1 else "0"
        return (dist.is_initialized() and dist.get_backend() == "nccl" and self.g.is_cuda
                and os.environ.get("MMRCA_GRAPH_DP", default) == "1")

    def span_ready(self, lo: int, hi: int, flush: bool = False):
        """Called by the engine when grads in [lo, hi) are final.  Adjacent ready spans are merged until a bucket is
        full (spans arrive in descending address order within an encoder)."""
        if not self.enabled or (self.world == 1 and not self.force) or hi <= lo:
            return
        if self._acc_lo is not None and hi == self._acc_lo:
            self._acc_lo = lo
        elif self._acc_lo is not None and lo == self._acc_hi:
            self._acc_hi = hi
        else:
            self._flush()
            self._acc_lo, self._acc_hi = lo, hi
        if flush or (self._acc_hi - self._acc_lo) >= self.bucket_elems:
            self._flush()

    def _flush(self):
        if self._acc_lo is None:
            return
        view = self.g[self._acc_lo:self._acc_hi]
        op = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM
        wire = view if self.wire_dtype is None else view.to(self.wire_dtype)       # cast on the stream that produced the span
        work = dist.all_reduce(wire, op=op, async_op=True)
        self.pending.append((work, view, op, wire))
        self.bytes_reduced += wire.numel() * wire.element_size()
        self.launches += 1
        self._acc_lo = self._acc_hi = None

    def finish(self):
        self._flush()
        timed = (self.timing and self.pending and torch.cuda.is_available() and self.g.is_cuda
                 and not torch.cuda.is_current_stream_capturing())
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, view, op, wire in self.pending:
            work.wait()
            if wire is not view:
                view.copy_(wire)
            if op == dist.ReduceOp.SUM:
                view.div_(self.world)
        if timed:
            e1.record()
            self.wait_events.append((e0, e1))
        self.pending.clear()


_HOST_GROUP = []


def host_group():
    """A process group for small HOST-side agreements between the ranks (a gloo group next to the RCCL one: all-reducing a CPU integer
    does not touch the GPU's queue, so the launch thread never waits for the step it has just enqueued).  Created on first use -- a
    collective call: every rank must reach it at the same point (run_one_epoch's first batch).  None with a single rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if not _HOST_GROUP:
        _HOST_GROUP.append(dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo"))
    return _HOST_GROUP[0]


def agree_caption_width(width: int) -> int:
    """max over the ranks of this batch's trimmed caption width (training.caption_width): the width is part of a HIP graph's key, and
    ranks that key differently would warm up, capture and replay at different steps.  A wider layout only adds masked key columns
    (same logits and gradients), so the maximum is safe for every rank."""
    g = host_group()
    if g is None:
        return int(width)
    t = torch.tensor([int(width)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=g)
    return int(t.item())


def broadcast_seed(seed: Optional[int], device="cpu") -> int:
    """One shuffle seed for every rank: the given one, or a random draw of rank 0 (the reference's loader is unseeded)."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([seed], dtype=torch.int64, device=device)
        dist.broadcast(t, src=0)
        seed = int(t.item())
    return seed


def all_reduce_counts(correct: int, count: int, device) -> tuple:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return correct, count
    t = torch.tensor([correct, count], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return int(t[0].item()), int(t[1].item())
