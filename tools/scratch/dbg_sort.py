import torch
m = torch.tensor([[1,0,0,1,0],[1,0,224,1,0],[1,224,0,0,0],[2,0,0,1,0],[3,0,0,1,0],[1,0,0,1,2],[1,224,0,1,2],[2,0,0,1,2],[3,0,0,1,2],[1,0,0,1,1],[1,224,0,1,1],[2,0,0,0,1],[2,224,0,1,1],[3,0,0,1,1]], dtype=torch.int32).cuda()
o = torch.argsort(m[:, 4], stable=True)
print(o.tolist())
print(m[o].tolist())
o2 = torch.argsort(m[:, 4].contiguous(), stable=True)
print(o2.tolist())
o3 = torch.sort(m[:, 4].contiguous().long(), stable=True)[1]
print(o3.tolist())
print(torch.argsort(m[:, 4].cpu(), stable=True).tolist())
