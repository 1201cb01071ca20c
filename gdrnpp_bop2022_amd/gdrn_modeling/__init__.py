"""Host-side mirror of the reference's ``core/gdrn_modeling`` inference surface:
``build_model_optimizer(cfg, is_test)``, ``GDRN_DoubleMask.forward`` and the evaluator-style
post-processing, with the per-ROI CPU / GL work replaced by calls into ``libgdrnpp_hip.so``."""
