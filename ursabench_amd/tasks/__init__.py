"""Task namespace, looked up by attribute name (`getattr(tasks, args.task)`,
URSABench/experiment.py:82) like URSABench/tasks/__init__.py:1-5."""
from .task_base import _Task  # noqa: F401
from .prediction import Prediction  # noqa: F401
from .ood_detection import OODDetection  # noqa: F401
from .decision_making import Decision  # noqa: F401
from .distilled import OODDetectionDistilled, PredictionDistilled  # noqa: F401
