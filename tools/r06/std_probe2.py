import sys
sys.path.insert(0, "on-policy_amd")
import numpy as np, torch
from onpolicy.algorithms.utils import fused_mlp
dev = torch.device("cuda", 0)
torch.manual_seed(0)
f32, f64 = np.float32, np.float64
def fma(a, b, c): return f32(f64(a) * f64(b) + f64(c))
for D in (48, 18):
    x = torch.randn(256, D, device=dev)
    ref = fused_mlp.standardize_rows(x).cpu().numpy()[:, :D]
    xn = x.cpu().numpy()
    def bfly(p):
        p = p.copy()
        for m in (8, 4, 2, 1):
            p = np.array([f32(p[i] + p[i ^ m]) for i in range(16)], f32)
        return p[0]
    def run(sum_mode, q_mode, var_mode, out_mode):
        bad = 0
        for r in range(256):
            row = xn[r]
            part = np.zeros(16, f32); 
            for sub in range(16):
                v = [row[4 * sub + e] if 4 * sub + e < D else f32(0) for e in range(4)]
                if sum_mode == 0: part[sub] = f32(f32(v[0] + v[1]) + f32(v[2] + v[3]))
                else: part[sub] = f32(f32(f32(v[0] + v[1]) + v[2]) + v[3])
            mean = f32(bfly(part) / f32(D))
            pq = np.zeros(16, f32)
            for sub in range(16):
                q = f32(0)
                for e in range(4):
                    if 4 * sub + e < D:
                        d = f32(row[4 * sub + e] - mean)
                        if q_mode == 0: q = fma(d, d, q)
                        elif q_mode == 1: q = f32(q + f32(d * d))
                pq[sub] = q
            qs = bfly(pq)
            if var_mode == 0: var = f32(f32(qs / f32(D)) + f32(1e-5))
            else: var = fma(qs, f32(f32(1) / f32(D)), f32(1e-5))
            rstd = f32(f32(1) / f32(np.sqrt(var)))
            if out_mode == 0: o = np.array([f32(f32(v - mean) * rstd) for v in row], f32)
            else: o = np.array([fma(v, rstd, f32(-f32(mean * rstd))) for v in row], f32)
            bad += int((o != ref[r]).sum())
        return bad
    for sm in (0, 1):
        for qm in (0, 1):
            for vm in (0, 1):
                for om in (0, 1):
                    print(D, "sum", sm, "q", qm, "var", vm, "out", om, "mismatches", run(sm, qm, vm, om))
