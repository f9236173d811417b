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


def test_graphed_forward_waits_for_vectors_still_on_the_meta_stream(dev, tmp_path):
    """model(x, metax, mask) in eval mode under no_grad -- the shape of train_meta.py's test() call --
    defers the wait for the reweighting vectors to their first reader; with inference_graphs that reader is the copy into
    the graph's static buffer, which has to wait for the "meta" side stream first."""
    from fewshot_detection_amd import cfgs, streams
    from fewshot_detection_amd.darknet_meta import Darknet
    assert streams.ENABLED and streams.META
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(3)
    net = Darknet(dyn_cfg, rw_cfg).to(dev).eval()
    x = torch.rand(2, 3, 160, 160, device=dev)
    # large supports: the reweighting net is still running on its stream when the head wants the vectors
    metas = [(torch.rand(8, 3, 416, 416, device=dev), (torch.rand(8, 1, 416, 416, device=dev) > 0.5).float()) for _ in range(4)]
    with torch.no_grad():
        eager = [net(x, mx, mk).clone() for mx, mk in metas]
        net.inference_graphs = True
        for rep in range(2):
            for (mx, mk), ref in zip(metas, eager):
                assert torch.equal(net(x, mx, mk), ref)


def test_eval_fold_is_rebuilt_after_a_training_forward_moved_the_running_statistics(dev, tmp_path):
    """eval -> train-mode forward (no optimizer step) -> eval.  The training pass rewrites running_mean /
    running_var through raw pointers; the folded eval weights (and captured graphs) must not be served stale."""
    from fewshot_detection_amd import cfgs, engine
    from fewshot_detection_amd.darknet_meta import Darknet
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(4)
    net = Darknet(dyn_cfg, rw_cfg).to(dev)
    x = torch.rand(2, 3, 96, 96, device=dev)
    vecs = [torch.rand(3, 1024, 1, 1, device=dev)]
    for graphs in (False, True):
        net.inference_graphs = graphs
        net.eval()
        with torch.no_grad():
            before = net.detect_forward(x, vecs).clone()
        net.train()
        with torch.no_grad():
            net.detect_forward(torch.rand(4, 3, 96, 96, device=dev) * 3.0, vecs)      # moves every running statistic
        net.eval()
        with torch.no_grad():
            after = net.detect_forward(x, vecs).clone()
            old = engine.FOLD_EVAL_BN
            engine.FOLD_EVAL_BN = False
            net.inference_graphs = False
            try:
                unfolded = net.detect_forward(x, vecs).clone()
            finally:
                engine.FOLD_EVAL_BN = old
        assert not torch.equal(after, before)
        assert float((after - unfolded).norm() / unfolded.norm()) < 2e-4
