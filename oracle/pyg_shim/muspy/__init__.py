"""Empty import stub (oracle only): utils.py:7 imports muspy at module scope."""
