"""Import-time stand-in (see Bio/__init__.py)."""
