import time, torch, numpy as np
n = 128 << 20
src = torch.empty(n, dtype=torch.uint8); src.fill_(1)
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True); pin.copy_(src)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
def rate(f, reps=8):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return n * reps / (time.perf_counter() - t) / 1e9
print(f"pageable H2D {rate(lambda: dst.copy_(src)):.1f} GB/s; pinned H2D {rate(lambda: dst.copy_(pin, non_blocking=True)):.1f} GB/s; host memcpy 1 thread {rate(lambda: pin.copy_(src)):.1f} GB/s", flush=True)
import threading
def par_copy(k):
    ths = [threading.Thread(target=lambda i=i: pin[i * n // k:(i + 1) * n // k].copy_(src[i * n // k:(i + 1) * n // k])) for i in range(k)]
    for t in ths: t.start()
    for t in ths: t.join()
torch.set_num_threads(1)
for k in (2, 4, 8):
    print(f"host memcpy {k} threads {rate(lambda: par_copy(k)):.1f} GB/s", flush=True)
