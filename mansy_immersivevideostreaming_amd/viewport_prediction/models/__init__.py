from .mtio import ViewportTransformerMTIO, FusedAdamW  # noqa: F401
