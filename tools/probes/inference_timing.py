import sys, time, random, numpy as np, torch
sys.path.insert(0, "/root/repo")
from fewshot_detection_amd import episode as E, utils
dev = torch.device("cuda:0")
rng = np.random.RandomState(0); random.seed(0)
imgs = [rng.randint(0, 256, (375, 500, 3)).astype(np.uint8) for _ in range(64)]
aug = E.DeviceAugmenter(dev)
def once():
    params = [E.draw_augmentation(500, 375) for _ in imgs]
    return aug(imgs, params, (416, 416), layout="nhwc4")
once(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): once()
torch.cuda.synchronize()
print("augment 64 x 500x375 -> 416x416 nhwc4 incl. host tables + H2D: %.2f ms per batch" % ((time.perf_counter() - t0) / 5 * 1e3))
from fewshot_detection_amd import ops
ops.kernel_profile(True); once(); torch.cuda.synchronize(); ops.kernel_profile(False)
print({k: v for k, v in ops.kernel_profile_collect().items() if v["launches"]})
# decode + NMS timing at B=64, N=15
out = torch.randn(64 * 15, 30, 13, 13, device=dev)
ANCH = [1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071]
torch.cuda.synchronize(); t0 = time.perf_counter()
boxes = utils.get_region_boxes_v2(out, 15, 0.005, 1, ANCH, 5, 0, 1)
t1 = time.perf_counter()
kept = [utils.nms(b, 0.45) for b in boxes]
t2 = time.perf_counter()
print("decode 960 rows: %.1f ms (device kernel + list building), nms of all rows: %.1f ms; boxes %d -> kept %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, sum(len(b) for b in boxes), sum(len(k) for k in kept)))
b0 = boxes[0]
t0 = time.perf_counter(); utils._nms_host([list(b) for b in b0], 0.45); t1 = time.perf_counter()
print("host python nms of ONE row (%d boxes): %.1f ms" % (len(b0), (t1 - t0) * 1e3))
