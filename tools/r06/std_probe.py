import sys, os
sys.path.insert(0, "on-policy_amd")
import numpy as np, torch
from onpolicy import _native
from onpolicy.algorithms.utils import fused_mlp
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for D in (18, 54, 48, 384):
    x = torch.randn(4096, D, device=dev)
    ref = fused_mlp.standardize_rows(x).cpu().numpy()[:, :D]
    lib = _native.lib()
    out = torch.zeros(4096, (D + 3) // 4 * 4, device=dev)
    raw = torch.zeros(4096 * D, device=dev)
    slab = (_native.Slab * 1)(_native.Slab(x.data_ptr(), raw.data_ptr(), x.numel()))
    std = (_native.StdSlab * 1)(_native.StdSlab(x.data_ptr(), out.data_ptr(), 4096, D, out.shape[1], 1e-5))
    assert lib.mappo_slab_copy_std(slab, 1, std, 1, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :D]
    xn = x.cpu().numpy()
    f32 = np.float32
    # emulate: lane sub handles k = 4 sub + 64 i
    def emu(fma_q, fma_out):
        res = np.zeros_like(xn)
        for r in range(64):
            row = xn[r]
            part = np.zeros(16, f32)
            for sub in range(16):
                s = f32(0)
                for k in range(4 * sub, D, 64):
                    v = [row[k + e] if k + e < D else f32(0) for e in range(4)]
                    s = f32(s + f32(f32(v[0] + v[1]) + f32(v[2] + v[3])))
                part[sub] = s
            def bfly(p):
                p = p.copy()
                for m in (8, 4, 2, 1):
                    p = np.array([f32(p[i] + p[i ^ m]) for i in range(16)], f32)
                return p[0]
            mean = f32(bfly(part) / f32(D))
            partq = np.zeros(16, f32)
            for sub in range(16):
                q = f32(0)
                for k in range(4 * sub, D, 64):
                    for e in range(4):
                        if k + e < D:
                            d = f32(row[k + e] - mean)
                            q = f32(np.float64(d) * np.float64(d) + np.float64(q)) if fma_q else f32(q + f32(d * d))
                partq[sub] = q
            var = f32(bfly(partq) / f32(D))
            rstd = f32(f32(1) / f32(np.sqrt(f32(var + f32(1e-5)))))
            res[r] = [(f32(f32(v - mean) * rstd)) for v in row]
        return res[:64]
    for fq in (True, False):
        e = emu(fq, False)
        print(D, "fma_q", fq, "emu==mlp", int((e != ref[:64]).sum()), "emu==copy", int((e != got[:64]).sum()), "mlp!=copy", int((ref[:64] != got[:64]).sum()))
