"""Empty import stub (oracle only): utils.py:8 imports PrettyTable at module scope."""


class PrettyTable:
    def __init__(self, *a, **k):
        raise NotImplementedError
