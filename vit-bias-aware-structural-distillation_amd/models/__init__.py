from .teacher import (TeacherModel, estimate_intrinsic_dim, extract_intermediates, load_teacher,  # noqa: F401
                      probe_model)
from .vit import VisionTransformer, create_vit, VIT_PRESETS  # noqa: F401
