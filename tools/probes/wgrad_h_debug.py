import sys, torch
sys.path.insert(0, "/root/repo")
from fewshot_detection_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
def view(x):
    B, C, H, W = x.shape
    return ops.View(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(dev).to(BF), B, H, W, C)
B, H, W, cin, cout, k = 1, 8, 8, 64, 128, 1
pix = B * H * W
ones_x = torch.ones(B, cin, H, W); ones_y = torch.ones(B, cout, H, W)
dw = ops.conv2d_wgrad(view(ones_y), cout, view(ones_x), cin, k).cpu()
print("ones/ones (expect %d everywhere): min %.1f max %.1f" % (pix, dw.min(), dw.max()))
chy = torch.arange(cout).float().view(1, cout, 1, 1).expand(B, cout, H, W)
dw = ops.conv2d_wgrad(view(chy.contiguous()), cout, view(ones_x), cin, k).cpu()[:, :, 0, 0]
print("dy=channel idx: dw[co, 0] / pix first 20:", (dw[:20, 0] / pix).tolist(), " ... row const over ci:", bool((dw == dw[:, :1]).all()))
chx = torch.arange(cin).float().view(1, cin, 1, 1).expand(B, cin, H, W)
dw = ops.conv2d_wgrad(view(ones_y), cout, view(chx.contiguous()), cin, k).cpu()[:, :, 0, 0]
print("x=channel idx: dw[0, ci] / pix first 20:", (dw[0, :20] / pix).tolist())
px = torch.arange(pix).float().view(B, 1, H, W)
dw = ops.conv2d_wgrad(view(px.expand(B, cout, H, W).contiguous()), cout, view(ones_x), cin, k).cpu()[:, :, 0, 0]
print("dy=pixel idx (expect %d): " % (pix * (pix - 1) // 2), dw[0, 0].item(), dw[5, 7].item())
dw = ops.conv2d_wgrad(view(px.expand(B, cout, H, W).contiguous()), cout, view(px.expand(B, cin, H, W).contiguous()), cin, k).cpu()[:, :, 0, 0]
print("dy=x=pixel idx (expect %d): " % sum(i * i for i in range(pix)), dw[0, 0].item(), dw[5, 7].item())
