"""Generated cfg text == the reference's cfg files, block by block (needs /root/reference; CPU only)."""
import os
import sys

import pytest

from fewshot_detection_amd import cfgs
from fewshot_detection_amd.cfg import parse_cfg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import ref_shim  # noqa: E402


def _norm(blocks):
    return [{k: str(v).replace(" ", "") for k, v in b.items()} for b in blocks]


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
def test_generated_cfgs_equal_reference_files(tmp_path):
    dyn, rw, tiny = cfgs.write_standard_cfgs(str(tmp_path))
    for mine, ref in ((dyn, "darknet_dynamic.cfg"), (rw, "reweighting_net.cfg"), (tiny, "tiny-yolo-voc.cfg")):
        a = _norm(parse_cfg(mine))
        b = _norm(parse_cfg(os.path.join(ref_shim.REF, "cfg", ref)))
        assert len(a) == len(b), ref
        for i, (x, y) in enumerate(zip(a, b)):
            assert x == y, (ref, i, x, y)


def test_cfg_parser_basics(tmp_path):
    p = tmp_path / "t.cfg"
    p.write_text("# comment\n[net]\nwidth = 32\n\n[convolutional]\nfilters=8\n[cost]\ntype=sse\n")
    blocks = parse_cfg(str(p))
    assert blocks[0] == {"type": "net", "width": "32"}
    assert blocks[1] == {"type": "convolutional", "batch_normalize": 0, "filters": "8"}
    assert blocks[2] == {"type": "cost", "_type": "sse"}
