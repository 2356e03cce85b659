from .mtio import ViewportTransformerMTIO, FusedAdamW  # noqa: F401
from .linear_regression import LinearRegression  # noqa: F401
