"""Model-level parity on the MI355X: the cfg-driven Darknet on HIP kernels vs (a) golden vectors minted
from the reference, (b) the CPU oracle on fresh seeded inputs.  Tolerance 1e-3 (north_star), fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _mini(dev):
    from fewshot_detection_amd.darknet_meta import Darknet
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net.load_weights(os.path.join(GOLD, "mini.weights"))
    return net.to(dev)


def test_meta_detector_forward_vs_reference_golden(dev):
    d = np.load(os.path.join(GOLD, "mini_forward.npz"))
    net = _mini(dev)
    assert int(net.seen) == 4242
    x, metax, mask = (torch.from_numpy(d[k]).to(dev) for k in ("x", "metax", "mask"))
    net.eval()
    with torch.no_grad():
        dyn = net.meta_forward(metax, mask)
        out = net.detect_forward(x, dyn)
    assert dyn[0].shape == (3, 64, 1, 1) and out.shape == (6, 30, 2, 2)
    assert np.abs(dyn[0].cpu().numpy() - d["dyn_eval"]).max() < TOL
    assert np.abs(out.cpu().numpy() - d["out_eval"]).max() < TOL
    net.train()
    with torch.no_grad():
        out = net(x, metax, mask)
    assert np.abs(out.cpu().numpy() - d["out_train"]).max() < TOL
    sd = net.state_dict()
    assert np.allclose(sd["models.0.bn1.running_mean"].cpu().numpy(), d["bn1_mean_after"], atol=1e-5)
    assert np.allclose(sd["models.0.bn1.running_var"].cpu().numpy(), d["bn1_var_after"], atol=1e-5)
    assert np.allclose(sd["learnet_models.10.bn6.running_var"].cpu().numpy(), d["lbn6_var_after"], atol=1e-5)


def test_weights_roundtrip_byte_exact(dev, tmp_path):
    net = _mini(dev)
    p = str(tmp_path / "rt.weights")
    net.save_weights(p)
    assert open(p, "rb").read() == open(os.path.join(GOLD, "mini.weights"), "rb").read()


def test_plain_yolo_vs_reference_golden(dev):
    from fewshot_detection_amd.darknet import Darknet
    d = np.load(os.path.join(GOLD, "mini_yolo_forward.npz"))
    net = Darknet(os.path.join(GOLD, "mini_tiny_yolo.cfg"))
    net.load_weights(os.path.join(GOLD, "mini_yolo.weights"))
    net = net.to(dev)
    x = torch.from_numpy(d["x"]).to(dev)
    net.eval()
    with torch.no_grad():
        assert np.abs(net(x).cpu().numpy() - d["out_eval"]).max() < TOL
    net.train()
    with torch.no_grad():
        assert np.abs(net(x).cpu().numpy() - d["out_train"]).max() < TOL


@pytest.mark.parametrize("S,Sm,B,N", [(104, 104, 2, 3), (96, 64, 3, 2)])
def test_mini_detector_vs_oracle_odd_sizes(dev, S, Sm, B, N):
    """13x13 -> 6x6 floor pooling (S=104) and non-default support size, against the CPU oracle."""
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle.net import OracleDarknet
    torch.manual_seed(S)
    cfgs = (os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    ora = OracleDarknet(*cfgs)
    for m in ora.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    net = Darknet(*cfgs)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev)
    x, metax = torch.rand(B, 3, S, S), torch.rand(N, 3, Sm, Sm)
    mask = (torch.rand(N, 1, Sm, Sm) > 0.5).float()
    ora.train(); net.train()
    with torch.no_grad():
        ref = ora(x, metax, mask)
        out = net(x.to(dev), metax.to(dev), mask.to(dev))
    assert out.shape == ref.shape
    assert float((out.cpu() - ref).abs().max()) < TOL
