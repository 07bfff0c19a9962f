"""Small host utilities (attribute dict used for `opt` / `var`, as the reference's EasyDict)."""


class edict(dict):
    """Nested attribute dictionary with the subset of EasyDict behaviour the reference relies on
    (attribute access, recursive conversion of dicts on assignment, update, pop)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, edict):
            v = edict(v)
        elif isinstance(v, (list, tuple)) and any(isinstance(x, dict) for x in v):
            v = type(v)(edict(x) if isinstance(x, dict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v
