import torch, time
x = torch.empty(int(7.77e9) // 4, dtype=torch.float32, device="cuda").zero_()
for _ in range(2): s = x.sum()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): s = x.sum()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("torch sum of 7.77 GB: %.3f ms = %.2f TB/s" % (dt * 1e3, 7.77e9 / dt / 1e12))
