"""Import-time stand-in (see Bio/__init__.py); trimming is upstream of the hot path."""
__version__ = "3.1"
