"""hipGraph replay of the eval-mode detect_forward (Darknet.inference_graphs): same bits as the eager launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_inference_equals_eager(dev, tmp_path, dtype):
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.darknet_meta import Darknet
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(2)
    net = Darknet(dyn_cfg, rw_cfg).to(dev).eval().set_compute_dtype(dtype)
    vecs = [torch.rand(5, 1024, 1, 1, device=dev)]
    xs = [torch.rand(2, 3, 160, 160, device=dev) for _ in range(3)] + [torch.rand(1, 3, 96, 128, device=dev)]
    with torch.no_grad():
        eager = [net.detect_forward(x, vecs).clone() for x in xs]
        net.inference_graphs = True
        for rep in range(2):                       # second round replays the cached graphs
            for x, ref in zip(xs, eager):
                out = net.detect_forward(x, vecs)
                assert torch.equal(out, ref)
        assert len(net._graphs) == 2               # one per input shape
        # new vectors re-use the captured graph (it reads a static copy); new weights after load_weights-style in-place
        # updates need a fresh capture: fresh results either way
        vecs2 = [torch.rand(5, 1024, 1, 1, device=dev)]
        net.inference_graphs = False
        ref2 = net.detect_forward(xs[0], vecs2).clone()
        net.inference_graphs = True
        assert torch.equal(net.detect_forward(xs[0], vecs2), ref2)
        assert len(net._graphs) == 2
        conv0 = net.models[0][0]
        conv0.weight.mul_(1.5)                     # bumps the parameter's version
        net.inference_graphs = False
        ref3 = net.detect_forward(xs[0], vecs2).clone()
        net.inference_graphs = True
        out3 = net.detect_forward(xs[0], vecs2)
        assert torch.equal(out3, ref3) and not torch.equal(ref3, ref2)
    # training mode and autograd calls never take the graph path
    net.train()
    out = net.detect_forward(xs[0], vecs)
    assert out.requires_grad
