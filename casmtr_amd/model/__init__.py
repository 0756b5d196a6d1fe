"""a plain-PyTorch CasMTR-4c around the MI355X hot path -- backbone, position encodings, transformer blocks,
up-sampling, fine matching on torch ops; QuadTreeAttention, CoarseMatching and CascadeMatching on this package's HIP kernels.
Module / parameter names follow the reference so that its checkpoints load unchanged (`matcher.` prefix stripped)."""
from .casmtr4c import CasMTR2c, CasMTR4c, outdoor_2c_config, outdoor_4c_config  # noqa: F401
from .indoor import CasMTRIndoor4c, indoor_4c_config  # noqa: F401,E402
