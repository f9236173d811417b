import sys, torch
sys.path.insert(0, "/root/repo")
from fewshot_detection_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
def view(x):
    B, C, H, W = x.shape
    return ops.View(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(dev).to(BF), B, H, W, C)
B, H, W, cin, cout, k = 1, 8, 8, 64, 128, 1
pix = 64
px = torch.arange(pix).float().view(B, 1, H, W)
onehot = torch.zeros(B, cin, H, W)
for p in range(pix):
    onehot[0, p, p // W, p % W] = 1
dw = ops.conv2d_wgrad(view(px.expand(B, cout, H, W).contiguous()), cout, view(onehot), cin, k).cpu()[:, :, 0, 0]
print("x one-hot(p==ci), dy=pix idx: dw[0, ci] should be ci:")
print([int(v) for v in dw[0].tolist()])
oh_y = torch.zeros(B, cout, H, W)
for p in range(pix):
    oh_y[0, p, p // W, p % W] = 1
dw = ops.conv2d_wgrad(view(oh_y), cout, view(px.expand(B, cin, H, W).contiguous()), cin, k).cpu()[:, :, 0, 0]
print("dy one-hot(p==co), x=pix idx: dw[co, 0] should be co (co<64):")
print([int(v) for v in dw[:64, 0].tolist()])
