import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_learn import _agent, _data
from safe_control_gym_amd.ppo import policy_loss_terms, value_loss_term
ag = _agent(12, 128, 2, 'tanh')
M, mb = int(sys.argv[1]) if len(sys.argv) > 1 else 524288, int(sys.argv[2]) if len(sys.argv) > 2 else 65536
data = _data(12, 2, M, ag)
F = ag._build_fused(data, mb)
print('n_wg', F['args'].n_workgroups)
idx = torch.randperm(M, device='cuda')[:mb]
F['idx'].copy_(idx.to(torch.int32))
ag._flat['g'].zero_()
ag._fused_grad(F)
torch.cuda.synchronize()
got = ag._flat['g'].clone()
batch = {k: v[idx] for k, v in data.items()}
ag._flat['g'].zero_()
pl, el, kl = policy_loss_terms(ag.ac, batch, ag.cfg.clip_param)
vl = value_loss_term(ag.ac, batch, ag.cfg.clip_param, False)
(pl + ag.cfg.entropy_coef * el).backward(); vl.backward()
ref = ag._flat['g'].clone()
off = 0
for prefix, mod in (('actor', ag.ac.actor), ('critic', ag.ac.critic)):
    for name, prm in mod.named_parameters():
        k = prm.numel()
        g, r = got[off:off + k], ref[off:off + k]
        print(f'{prefix}.{name:28s} n={k:6d} max|ref|={r.abs().max().item():.3e} max|err|={(g - r).abs().max().item():.3e}', (g[:3].tolist(), r[:3].tolist()) if k <= 4 else '')
        off += k
print('kl', got[-1].item(), kl.item())
