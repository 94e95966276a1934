import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")   # 512 MiB
x.normal_()
torch.cuda.synchronize()
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
