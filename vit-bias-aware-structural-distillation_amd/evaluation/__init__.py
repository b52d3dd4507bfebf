from .metrics import evaluate_model, measure_efficiency, run_eval_suite, save_metrics  # noqa: F401
