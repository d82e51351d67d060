import sys, os, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "nvalchemi-toolkit-ops_amd")]
from tests.test_reference_scenarios_gpu import _simple_system, DEV
from nvalchemiops.interactions.electrostatics import pme_reciprocal_space
p, q, c = _simple_system(6, torch.float64)
kw = dict(alpha=0.3, mesh_dimensions=(16, 16, 16))
e, f = pme_reciprocal_space(p, q, c, compute_forces=True, **kw)
bi = torch.zeros(6, dtype=torch.int32, device=DEV)
eb, fb = pme_reciprocal_space(p, q, c.unsqueeze(0), batch_idx=bi, compute_forces=True, **kw)
print("single vs batch-of-one: dE", float((e - eb).abs().max()), "dF", float((f - fb).abs().max()), "F scale", float(f.abs().max()))
e2, f2 = pme_reciprocal_space(p, q, c, compute_forces=True, **kw)
print("repeat: dE", float((e - e2).abs().max()), "dF", float((f - f2).abs().max()))
pg = p.clone().requires_grad_(True)
eg = pme_reciprocal_space(pg, q, c, **kw)
print("composed vs fused dE", float((eg.detach() - e).abs().max()))
