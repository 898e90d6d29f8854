"""Data-parallel helpers: one process per GPU, env shards per rank, one flat-bucket all-reduce per
optimiser step over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; this module implements SURVEY §8e:
messages are tiny (PPO quadrotor-2D: 36 741 fp32 ≈ 147 KB), i.e. latency-bound on xGMI, so the number
of collectives is minimised — gradients of BOTH networks, the approx-KL of the actor gate and any other
scalar that every rank must agree on travel in ONE all-reduce.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            # SCG_DIST_BACKEND=gloo exercises the multi-rank control flow on a box with fewer GPUs than ranks
            backend = os.environ.get('SCG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device('cuda', local))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local % torch.cuda.device_count())
            dist.init_process_group(backend)
    return rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_capturable():
    """True when a collective on a device tensor can be recorded into a HIP graph with the kernels around it: RCCL (torch's "nccl"
    backend).  gloo stages through the host and cannot."""
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'


def allreduce_probe(nbytes_list=(147 * 1024, 246 * 1024), reps=200, device=None):
    """Latency of an in-place SUM all-reduce of `nbytes` fp32 on THIS process group, eager (one enqueue per call) and captured
    (50 collectives per HIP-graph replay), in microseconds per collective.  With one rank it measures the fixed cost of the RCCL
    path (enqueue, kernel launch, no wire) — the part of a data-parallel optimiser step that the number of GPUs does not change."""
    out = {}
    if not collectives_capturable():
        return {'error': 'needs an initialised "nccl" (RCCL) process group'}
    dev = device or torch.device('cuda', torch.cuda.current_device())
    for nbytes in nbytes_list:
        t = torch.zeros(nbytes // 4, device=dev)
        for _ in range(10):
            dist.all_reduce(t)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dist.all_reduce(t)
        e1.record()
        torch.cuda.synchronize(dev)
        row = {'eager_us': 1e3 * e0.elapsed_time(e1) / reps}
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(50):
                    dist.all_reduce(t)
            g.replay()
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(max(1, reps // 50)):
                g.replay()
            e1.record()
            torch.cuda.synchronize(dev)
            row['captured_us'] = 1e3 * e0.elapsed_time(e1) / (50 * max(1, reps // 50))
        except Exception as exc:                                    # noqa: BLE001
            row['captured_us'] = None
            row['capture_error'] = repr(exc)[:200]
        out[str(nbytes)] = row
    out['world'] = dist.get_world_size()
    return out


def _through_host(t):
    """gloo (tests: several ranks sharing one GPU) has no device collectives in this build: stage device tensors on the host."""
    return t.is_cuda and dist.get_backend() == 'gloo'


def _all_reduce(t, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _through_host(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def broadcast_(t, src=0):
    if world_size() > 1:
        if _through_host(t):
            h = t.cpu()
            dist.broadcast(h, src)
            t.copy_(h)
        else:
            dist.broadcast(t, src)
    return t


class FlatBucket:
    """Gradients of a fixed parameter list + a few scalars packed into one contiguous fp32 buffer."""

    def __init__(self, params, n_scalars=0):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        self.n_scalars = n_scalars
        dev = self.params[0].device
        self.buf = torch.zeros(sum(self.sizes) + n_scalars, dtype=torch.float32, device=dev)

    def pack(self, scalars=()):
        off = 0
        for p, n in zip(self.params, self.sizes):
            g = p.grad
            if g is None:
                self.buf[off:off + n].zero_()
            else:
                self.buf[off:off + n].copy_(g.reshape(-1))
            off += n
        for k, s in enumerate(scalars):
            self.buf[off + k] = s
        return self.buf

    def all_reduce_mean(self):
        w = world_size()
        if w > 1:
            _all_reduce(self.buf)
            self.buf.div_(w)
        return self.buf

    def unpack(self):
        off = 0
        for p, n in zip(self.params, self.sizes):
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.buf[off:off + n].view_as(p))
            off += n
        return self.buf[off:off + self.n_scalars]


def all_reduce_sum_(t):
    if world_size() > 1:
        _all_reduce(t)
    return t


def broadcast_parameters(modules, src=0):
    """Make every rank start from rank `src`'s weights."""
    if world_size() > 1:
        for m in modules:
            for p in m.parameters():
                broadcast_(p.data, src)
            for b in m.buffers():
                broadcast_(b.data, src)
