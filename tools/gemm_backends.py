"""Times the skinny GEMM shapes of the MAPPO update under both BLAS back ends of PyTorch-ROCm."""
import torch, time
dev = torch.device("cuda", 0)
shapes = [("critic L1 fwd", 2621440, 384, 64), ("hidden fwd", 2621440, 64, 64), ("actor L1 fwd", 2621440, 48, 64),
          ("gru gh", 262144, 64, 192), ("gru dh", 262144, 192, 64), ("head", 2621440, 64, 5)]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    print("==", lib, torch.backends.cuda.preferred_blas_library())
    for name, M, K, N in shapes:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        t1 = bench(lambda: torch.nn.functional.linear(a, w, b))
        t2 = bench(lambda: torch.mm(a, w.t()))
        g = torch.randn(M, N, device=dev)
        t3 = bench(lambda: torch.mm(g, w))          # dX
        S = M // 4096
        t4 = bench(lambda: torch.bmm(g.view(S, 4096, N).transpose(1, 2), a.view(S, 4096, K)).sum(0))   # split-K dW
        gb = (M * K + M * N) * 4 / 1e3
        print("%-14s M=%8d K=%4d N=%4d  linear %7.1f us  mm %7.1f us  dX %7.1f us  dW(splitK) %7.1f us   [fwd HBM floor %6.1f us @5TB/s]"
              % (name, M, K, N, t1, t2, t3, t4, gb / 5e3))
