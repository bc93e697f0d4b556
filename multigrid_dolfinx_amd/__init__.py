"""MI355X-native geometric multigrid V-cycle behind the call surface of the
reference's `multigrid.py`."""
__version__ = "0.1.0"
