from . import qat, quantizable  # noqa: F401
