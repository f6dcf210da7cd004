from .metrics import MetricsLogger, StepTimer, format_display_line  # noqa: F401
