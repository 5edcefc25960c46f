"""Plugin module for ``core_method: heter_model_baseline_w_gencomm_stage2`` (resolved by opencood/tools/train_utils.py:269-287: the first
attribute whose lower-cased name equals the module name without underscores). See INTEGRATION.md for the
two-line shim that exposes it as ``opencood.models.heter_model_baseline_w_gencomm_stage2``."""
from .heter_model import HeterModelBaselineWDiffCommStage2


class HeterModelBaselineWGenCommStage2(HeterModelBaselineWDiffCommStage2):
    pass
__all__ = ["HeterModelBaselineWGenCommStage2", "HeterModelBaselineWDiffCommStage2"]
