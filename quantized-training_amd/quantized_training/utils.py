"""Logging set-up decorator for driver scripts (upstream src/quantized_training/utils.py:75-144).

Weights & Biases sweeps and Slurm script generation are outside this engine; ``--project`` /
``--sweep_*`` are accepted by the parser and ignored with a warning unless ``wandb`` is importable.
"""
import datetime
import logging
import os
from functools import wraps
from pprint import pformat

__all__ = ["setup_logging", "SLURM_ARGS"]

logger = logging.getLogger(__name__)

# sub-command flags kept so that reference command lines still parse
SLURM_ARGS = {
    "job-name": {"type": str, "default": "test"},
    "partition": {"type": str, "default": "gpu"},
    "nodes": {"type": int, "default": 1},
    "time": {"type": str, "default": "48:00:00"},
    "gpus": {"type": str, "default": "1"},
    "cpus": {"type": int, "default": 8},
    "mem": {"type": str, "default": "16GB"},
    "output": {"type": str, "default": None},
    "error": {"type": str, "default": None},
    "exclude": {"type": str, "default": None},
    "nodelist": {"type": str, "default": None},
}


def setup_logging(func):
    @wraps(func)
    def wrapper(args, *fargs, **fkwargs):
        if getattr(args, "log_file", None) == "datetime":
            args.log_file = f"logs/{datetime.datetime.now():%Y-%m-%d_%H-%M-%S}.log"
        if getattr(args, "log_file", None):
            d = os.path.dirname(args.log_file)
            if d:
                os.makedirs(d, exist_ok=True)
        logging.basicConfig(
            filename=getattr(args, "log_file", None),
            format="%(asctime)s - %(levelname)s - %(name)s - %(message)s",
            datefmt="%m/%d/%Y %H:%M:%S",
            level=getattr(logging, getattr(args, "log_level", "WARNING")),
        )
        if getattr(args, "project", None) or getattr(args, "sweep_config", None) or getattr(args, "sweep_id", None):
            try:
                import wandb  # noqa: F401
                wandb.init(project=args.project, name=args.run_name, id=args.run_id, resume="allow")
            except ImportError:
                logger.warning("wandb is not installed: --project/--sweep_* are ignored")
        if getattr(args, "action", None) is not None:
            logger.warning("sub-command '%s' (script generation) is not supported by this engine", args.action)
            return None
        logger.info("Training/evaluation parameters: %s", pformat(vars(args)))
        return func(args, *fargs, **fkwargs)

    return wrapper
