import sys, tempfile, torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import cfgs, engine
from fewshot_detection_amd.darknet_meta import Darknet
dev = torch.device("cuda:0")
d = tempfile.mkdtemp(); dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(d)
torch.manual_seed(2)
net = Darknet(dyn_cfg, rw_cfg).to(dev).eval()
vecs = [torch.rand(5, 1024, 1, 1, device=dev)]
xs = [torch.rand(2, 3, 160, 160, device=dev) for _ in range(3)] + [torch.rand(1, 3, 96, 128, device=dev)]
with torch.no_grad():
    e1 = [net.detect_forward(x, vecs).clone() for x in xs]
    e2 = [net.detect_forward(x, vecs).clone() for x in xs]
    print("eager vs eager:", [float((a - b).abs().max()) for a, b in zip(e1, e2)])
    net.inference_graphs = True
    for rep in range(2):
        g = [net.detect_forward(x, vecs).clone() for x in xs]
        print("graph rep", rep, [float((a - b).abs().max()) for a, b in zip(e1, g)])
    engine.FOLD_EVAL_BN = False
    net.inference_graphs = False
    u = [net.detect_forward(x, vecs).clone() for x in xs]
    print("unfolded vs folded:", [float((a - b).abs().max()) for a, b in zip(e1, u)], float(e1[0].abs().max()))
