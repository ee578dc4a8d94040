"""Hydra `_target_` namespace, mirroring image2layout/train/models/generator.py:1-9 for the two
classes on the hot path:  generator._target_=ralf_amd.models.generator.<Class>  (the class names keep
the substrings the Trainer dispatches on: "RetrievalAugmented", "AuxilaryTask", "Autoreg")."""
from .ralf import (  # noqa: F401
    ConcateAuxilaryTaskAutoreg,
    ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg,
)
