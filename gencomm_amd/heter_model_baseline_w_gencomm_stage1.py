"""Plugin module for ``core_method: heter_model_baseline_w_gencomm_stage1`` (resolved by opencood/tools/train_utils.py:269-287: the first
attribute whose lower-cased name equals the module name without underscores). See INTEGRATION.md for the
two-line shim that exposes it as ``opencood.models.heter_model_baseline_w_gencomm_stage1``."""
from .heter_model import HeterModelBaselineWGenComm


class HeterModelBaselineWGenCommStage1(HeterModelBaselineWGenComm):
    pass
__all__ = ["HeterModelBaselineWGenCommStage1", "HeterModelBaselineWGenComm"]
