from .utils import (  # noqa: F401
    cosine_scheduler, get_params_groups, has_batchnorms, fix_random_seeds,
    compute_gradient_stats, cancel_gradients_last_layer, clip_gradients, grad_stats_async,
    resume_from_checkpoint,
)
