import torch, time
x = torch.empty(2_700_000_000 // 8, dtype=torch.int64, device="cuda")
for fn, name, nbytes in ((lambda: x.fill_(7), "fill 2.7 GB", 2.7e9), (lambda: x.zero_(), "zero 2.7 GB", 2.7e9)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(name, "%.3f ms  %.2f TB/s" % (ms, nbytes / ms / 1e9))
y = torch.empty(750_000_000 // 8, dtype=torch.int64, device="cuda")
s = y.sum()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): s = y.sum()
b.record(); torch.cuda.synchronize()
print("read 0.75 GB (sum) %.3f ms %.2f TB/s" % (a.elapsed_time(b) / 10, 0.75e9 / (a.elapsed_time(b) / 10) / 1e9))
