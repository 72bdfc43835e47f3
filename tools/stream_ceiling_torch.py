"""torch.Tensor.copy_ / clone / in-place xor_ on 2 GiB tensors: the framework's own elementwise kernels as one more
independent reference point next to tools/stream_ceiling.hip."""
import torch, sys

def t(name, nbytes, f, reps=20):
    f(); f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name:44s} {ts[0]:8.3f} ms best {ts[len(ts)//2]:8.3f} ms median   {nbytes/ts[0]/1e9:6.2f} TB/s best {nbytes/ts[len(ts)//2]/1e9:6.2f} TB/s median", flush=True)

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048) << 20
a = torch.ones(n // 8, dtype=torch.int64, device="cuda")
b = torch.zeros_like(a)
print(f"-- (7) torch {torch.__version__}, tensors of {n/2**30:.2f} GiB")
t("torch b.copy_(a)", 2.0 * n, lambda: b.copy_(a))
t("torch a.bitwise_xor_(5)  (in place)", 2.0 * n, lambda: a.bitwise_xor_(5))
t("torch torch.bitwise_xor(a, 5, out=b)", 2.0 * n, lambda: torch.bitwise_xor(a, 5, out=b))
af = a.view(torch.float32); bf = b.view(torch.float32)
t("torch bf.copy_(af) float32", 2.0 * n, lambda: bf.copy_(af))
t("torch a.zero_()", 1.0 * n, lambda: a.zero_())
