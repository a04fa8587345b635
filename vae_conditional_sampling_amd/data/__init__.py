from .missing_data_module import MissingDataModule  # noqa: F401
