"""Pure-torch restatement of the PyG 2.0.2 symbols the reference imports.
Oracle infrastructure only — see ../README.md."""
__version__ = "2.0.2-shim"
