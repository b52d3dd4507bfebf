from .combined import BASDLoss, _align_token_count  # noqa: F401
from .layer_selector import GrassmannianLayerSelector, marchenko_pastur_rank  # noqa: F401
from .relational import geometric_relational_loss  # noqa: F401
