#!/usr/bin/env python3
"""A/B of the fused SAC gradient step between builds of csrc/scg_sac.hip (GPU box): for each library given — the shipped one and
tagged variants `libscg_sac_24_128_4_relu_<tag>.so` built here from another copy of the source — (a) a SHA-256 of the agent's flat
parameter / target / Adam vectors after 5 x 16 gradient steps from one seed (equal hashes = bit-identical steps), (b) microseconds per
gradient step from HIP events around replays of a 16-step graph, variants alternating.

    python tools/sac_step_ab.py [tag ...]          ('' = the shipped library; default: '' base)
"""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(tag):
    """The shipped library through the product loader; a tagged variant straight through ctypes (it may predate entry points the
    product loader binds, e.g. scg_sac_update_n)."""
    from safe_control_gym_amd import _sac
    _sac._libs.pop((24, 128, 4, 'relu'), None)
    if not tag:
        return _sac.lib(24, 128, 4, 'relu')
    D = C.CDLL(_sac.lib_path(24, 128, 4, 'relu')[:-3] + f'_{tag}.so')
    D.scg_sac_last_error.restype = C.c_char_p
    D.scg_sac_workspace_bytes.restype = C.c_size_t
    D.scg_sac_workspace_bytes.argtypes = [C.c_int]
    D.scg_sac_update.argtypes = [C.POINTER(_sac.SacArgs), C.c_void_p]
    return D


class Run:
    def __init__(self, tag, torch, steps=16):
        from safe_control_gym_amd import _sac
        from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
        self.torch, self.tag, self.steps = torch, tag, steps
        dev = torch.device('cuda', 0)
        torch.manual_seed(11)
        low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)
        self.ag = ag = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu'), dev)
        self.buf = buf = DeviceReplay(1_000_000, 24, 4, dev)
        g = torch.Generator(device=dev).manual_seed(2)
        n = 500_000
        obs = torch.randn(n, 24, device=dev, generator=g)
        act = torch.rand(n, 4, device=dev, generator=g) * 2 - 1
        rew = -((act - torch.tanh(obs[:, :4])) ** 2).sum(-1)
        buf.push(obs, act, rew, torch.randn(n, 24, device=dev, generator=g), (torch.rand(n, device=dev, generator=g) < 0.9).float())
        D = load(tag)
        _sac._libs[(24, 128, 4, 'relu')] = D                     # _fused_args asks _sac.lib for it
        self.F = F = ag._fused_args(buf, 4096)
        self.D = D
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        F['args'].phases = 0
        self.has_n = hasattr(D, 'scg_sac_update_n')
        if self.has_n:
            D.scg_sac_update_n.argtypes = [C.POINTER(_sac.SacArgs), C.c_int, C.c_void_p]
        self.graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            with torch.cuda.graph(self.graph, stream=s):
                st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                if self.has_n:
                    _sac.check(D, D.scg_sac_update_n(C.byref(F['args']), steps, st))
                else:
                    for _ in range(steps):
                        _sac.check(D, D.scg_sac_update(C.byref(F['args']), st))
        torch.cuda.current_stream(dev).wait_stream(s)

    def digest(self):
        self.torch.cuda.synchronize()
        h = hashlib.sha256()
        fl = self.ag._flat
        for k in ('p', 'targ', 'm', 'v', 'steps', 'counter'):
            h.update(fl[k].detach().cpu().numpy().tobytes())
        h.update(self.F['stats'].cpu().numpy().tobytes())
        return h.hexdigest()[:16]

    def time(self, replays=60):
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.graph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(replays):
            self.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / (replays * self.steps)


def main():
    import torch
    tags = sys.argv[1:] or ['', 'base']
    tags = ['' if t in ('shipped', '-') else t for t in tags]
    runs = []
    for t in tags:
        r = Run(t, torch)
        for _ in range(5):
            r.graph.replay()
        print(f'{t or "shipped":10s} update_n={r.has_n}  state hash after 80 steps: {r.digest()}  losses {[round(float(x), 6) for x in r.F["stats"].tolist()]}', flush=True)
        runs.append(r)
    for rnd in range(3):
        print('  '.join(f'{r.tag or "shipped"}: {r.time():7.2f} us/step' for r in runs), flush=True)


if __name__ == '__main__':
    main()
