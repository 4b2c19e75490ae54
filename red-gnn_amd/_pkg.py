"""Top-level names of the package (re-exported by the ``red_gnn_amd`` alias module)."""
__version__ = "0.1.0"
__all__ = ["__version__"]
