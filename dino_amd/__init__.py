"""dino_amd: MI355X-native DINO self-distillation training step (HIP kernels behind the
reference's Python API).  See DESIGN.md / INTEGRATION.md."""
from .config import get_default_dtype, set_default_dtype  # noqa: F401
