"""Stub modules so the reference (/root/reference) can be imported in THIS container to
generate golden vectors.  Only used by tools/gen_golden_*.py; never shipped to the GPU box
as a dependency (the reference does not exist there).

Stubs: munch.Munch (attr-dict), gym.Env + gym.spaces.Discrete, prettytable.PrettyTable.
"""
import sys
import types


def install():
    if 'munch' not in sys.modules:
        m = types.ModuleType('munch')

        class Munch(dict):
            __getattr__ = dict.__getitem__
            __setattr__ = dict.__setitem__
        m.Munch = Munch
        sys.modules['munch'] = m
    if 'gym' not in sys.modules:
        g = types.ModuleType('gym')
        sp = types.ModuleType('gym.spaces')

        class Env:
            def __init__(self, *a, **k):
                pass

        class Discrete:
            def __init__(self, n):
                self.n = n
        g.Env = Env
        sp.Discrete = Discrete
        g.spaces = sp
        sys.modules['gym'] = g
        sys.modules['gym.spaces'] = sp
    if 'prettytable' not in sys.modules:
        p = types.ModuleType('prettytable')

        class PrettyTable:
            def __init__(self, *a, **k):
                self.field_names = []
                self.rows = []

            def add_row(self, r):
                self.rows.append(r)

            def __str__(self):
                return '\n'.join(str(r) for r in self.rows)
        p.PrettyTable = PrettyTable
        sys.modules['prettytable'] = p


def legacy_transformer_signature():
    """Context manager reproducing torch<=2.0's nn.Transformer.__init__ positional signature
    (no `bias` parameter), so customized_transformer.py:46-49 builds the with-bias layout that
    the reference's documented environment (README: torch<=2.0) produces."""
    import contextlib
    import torch.nn as nn

    @contextlib.contextmanager
    def cm():
        orig = nn.Transformer.__init__

        def patched(self, *args, **kwargs):
            if len(args) == 14 and not kwargs:        # (..., norm_first, device, dtype)
                *head, device, dtype = args
                return orig(self, *head, True, device, dtype)
            return orig(self, *args, **kwargs)
        nn.Transformer.__init__ = patched
        try:
            yield
        finally:
            nn.Transformer.__init__ = orig
    return cm()
