from .data_parallel import GradientReducer  # noqa: F401
from .optim import AdamWScheduleFree, FlatParams  # noqa: F401
from .trainer import Trainer, _extract_student  # noqa: F401
