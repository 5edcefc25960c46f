"""The hypes_yaml config surface, swept: tests/golden/yaml_sweep.json is the report of oracle/sweep_yamls.py, which constructs this
build's model shell from every yaml the reference ships under opencood/hypes_yaml/**/GenComm_yamls/** (101 files) and records
the checkpoint-key count + a hash of the sorted (key, shape) pairs, or the reason a shell cannot be built. CPU only.

Where the reference checkout is mounted the sweep is re-run and must reproduce the committed report; elsewhere (the GPU box)
the report itself is checked: every GenComm yaml is either built or refused with a reason that names the yaml key."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPORT = os.path.join(HERE, "golden", "yaml_sweep.json")


def _report():
    with open(REPORT) as f:
        return json.load(f)


def test_report_covers_every_shipped_gencomm_yaml_and_names_every_refusal():
    rep = _report()
    assert len(rep) == 101
    gencomm = {k: v for k, v in rep.items() if "/gencomm/" in k}
    assert len(gencomm) == 28                                     # 28 shipped GenComm yamls
    built = {k: v for k, v in gencomm.items() if v["built"]}
    assert len(built) == 18
    for k, v in gencomm.items():
        assert v["core_method"].startswith("heter_model_baseline_w_gencomm"), k
        if v["built"]:
            assert v["keys"] > 200 and len(v["hash"]) == 16
        else:
            # camera (LSS) agents: SURVEY.md section 2 row 9, out of scope; two yamls carry `enhancer: enhancev12`, a string the
            # reference's own constructor fails on (stage2.py:153)
            assert ("lift_splat_shoot" in v["reason"] and v["reason"].startswith("NotImplementedError")) or \
                   ("model.args.enhancer" in v["reason"] and v["reason"].startswith("TypeError")), (k, v["reason"])
    # every yaml outside the gencomm/ folders names a baseline method's shell (MPDA, BackAlign, CodeFilling, STAMP, plain heter models)
    for k, v in rep.items():
        if "/gencomm/" not in k:
            assert not v["built"] and v["reason"].startswith("not a GenComm model"), (k, v)
    # identical configurations (OPV2V and DAIR-V2X share the m1 stage-1 model) hash identically
    assert rep["opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml"]["hash"] == rep["dairv2x/GenComm_yamls/gencomm/stage1/m1_att.yaml"]["hash"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/opencood/hypes_yaml"), reason="the reference checkout is not mounted here")
def test_sweep_reproduces_the_committed_report():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import sweep_yamls
    assert sweep_yamls.sweep() == _report()
