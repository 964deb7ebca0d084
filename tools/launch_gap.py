import torch, time
x = torch.zeros(64, device="cuda")
for _ in range(100): x.add_(1)
torch.cuda.synchronize()
n=5000
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
t=time.perf_counter(); e0.record()
for _ in range(n): x.add_(1)
th=(time.perf_counter()-t)/n*1e6
e1.record(); torch.cuda.synchronize()
print(f"tiny torch kernel: host {th:.2f} us per launch, device {e0.elapsed_time(e1)/n*1e3:.2f} us per launch")
