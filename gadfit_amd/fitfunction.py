"""fitfunc -- the abstract fitting function (mirror of fortran/gadfit/fitfunction.F90:32-64).

A model extends ``fitfunc`` and provides ``init`` (allocate parameters, optionally name
them) and ``eval(x) -> advar`` written with advar operators on ``self.pars``.  Parameter
indices are 1-based in ``set`` / ``get_index`` / ``get_name`` like the reference;
``self.pars`` itself is a Python list.
"""
from . import ad


class Par:
    """Host-side parameter slot: value + name (the public ``val`` of pars(:), AD:65-68)."""
    __slots__ = ('val', 'name')

    def __init__(self, val=0.0, name=None):
        self.val = float(val)
        self.name = name


class fitfunc:
    pars = None

    def init(self):  # deferred, fitfunction.F90:47-53
        raise NotImplementedError

    def eval(self, x):  # deferred, fitfunction.F90:59-63
        raise NotImplementedError

    def allocate(self, n):
        """allocate(this%pars(n)) in the reference's init procedures."""
        self.pars = [Par() for _ in range(n)]

    @property
    def parnames(self):
        return [p.name for p in self.pars]

    def set(self, par, value):
        """set(index, value) | set(name, value) | set(index, name) (fitfunction.F90:66-109)."""
        if isinstance(value, str):
            self.pars[int(par) - 1].name = value
        else:
            i = self.get_index(par) if isinstance(par, str) else int(par)
            self.pars[i - 1].val = float(value)

    def get_index(self, name):
        for i, p in enumerate(self.pars):
            if p.name == name:
                return i + 1
        raise KeyError('Parameter with name \'%s\' not found.' % name)   # fitfunction.F90:121-125

    def get_name(self, index):
        return self.pars[index - 1].name

    def trace(self):
        """Record eval() into a model tape (replaces per-point dynamic dispatch)."""
        saved = self.pars

        def fn(p, x):
            self.pars = p
            try:
                return self.eval(x)
            finally:
                self.pars = saved
        return ad.trace_model(fn, len(saved))

    def trace_variants(self, configure=None):
        """An eval() that compares AD variables (automatic_differentiation.F90:315-395) cannot be recorded symbolically: the empty
        set of its recorded paths (tape.Variants), to be filled by recordings at concrete points (gadfit.py: over the data)."""
        from . import tape as T
        saved = self.pars

        def fn(p, x):
            self.pars = p
            try:
                return self.eval(x)
            finally:
                self.pars = saved
        return T.Variants(fn, len(saved), configure=configure)
