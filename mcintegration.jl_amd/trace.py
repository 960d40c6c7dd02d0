"""A Python closure as integrand at device speed: the closure is run ONCE on symbolic draws and what it computes is written out as
the HIP C++ body the sample-batch kernels are JIT-compiled around (integrand.Integrand) -- the counterpart of Julia inlining
`integrand(var, config)` into the reference's loop (vegas/montecarlo.jl:140-144, vegas_mc/updates.jl:67-75, mcmc/montecarlo.jl:34-36).

    f = lambda x, c: np.exp(-np.sum(x * x) / 2) / (2 * np.pi) ** (len(x) / 2)
    integrate(f, var=Continuous(-5, 5), dof=[[4]], solver="vegas")        # (tracing is the default; or Integrand = trace_integrand(f, config))

The closure sees what a host closure sees (integrand.HostIntegrand) without the batch axis: with one variable type `x[i]` is the
i-th draw, with several `x` is a tuple with one array per variable type (a CompositeVar is indexed [leaf][slot] like the reference's
-- `x, y = cvar` -- a FermiK pool [slot][component]); the arrays are numpy object arrays of `Sym` nodes, so indexing, slicing,
arithmetic, `sum`, `np.sum / np.prod / np.dot` and the elementwise numpy functions (`np.exp`, `np.log`, `np.sqrt`, `np.sin`, ... --
numpy calls the method of the same name on an object) work as they are.  Python branches on sampled values (`1.0 if x[0] ** 2 + x[1] **
2 < 1 else 0.0`, `if` / `elif`, `and` / `or`) are written out as selects: the closure is run once per way through them (explore(); at
most MAX_WAYS ways).  A float array reached through `config.userdata` (the array itself, an attribute of a struct of parameters, an
entry of a dict) or captured by the closure may be indexed with a sampled value -- `grid[bin[0] - 1]`, `para.extQ[ext[0] - 1]`, the
reference's histogram examples (docs/src/index.md "Measure Histogram", test/bubble.jl:60): the table goes into the userdata vector
and the element is `ud[base + (int)index]` (_Table); a measure may add to a sampled bin, `obs[i][bin[0] - 1] += weights[i]` (_Obs).
Complex weights are pairs of real expressions (CSym).  What cannot be written out raises TraceError and the caller falls back to the
host batch-callback path: `math.*` functions (they want a float), `np.where / np.maximum` on ARRAYS of symbols (numpy compares and
truth-tests the elements itself; on single draws they trace, and `mci.trace.where / fmax / fmin` take arrays too), a Python list
indexed with a draw.  The written-out body is checked against the closure itself at random points of the domain before it is used
(the closure is called with plain arrays, one sample at a time; Discrete draws are integers there like on the host path and in the
reference): a closure that is not a pure function of its draws (hidden state, a branch taken on something the trace did not see) is
refused.

Captured parameters.  Floats the closure captures -- closure cells, float defaults, module-level floats its code names, floats read
off `config.userdata` (a number, an attribute of a struct of parameters, a dict entry); float arrays of up to 64 elements likewise --
are traced as PARAMETERS, not as literals: every maximal subexpression that depends on parameters
and constants only is evaluated on the host and handed to the kernel in a `ud[k]` slot (Integrand.userdata), so the written-out body is
the same text for every value and a parameter sweep over a closure reuses ONE code object (the kernel cache is keyed by the source).
The helper functions the closure reaches (Python functions in its cells, functions of its own module that its code names) are treated
the same way, so a table or a temperature captured by `green(tau, omega, beta)` next to the integrand is a parameter too.
Captured ints stay literals (they are structure more often than data: `range(n)`, `x[:n]`, `** n`); a closure that branches on a
captured float is traced again with its captured values as literals."""
import math
import types

import numpy as np

from .integrand import Integrand, Measure


class TraceError(Exception):
    """the closure cannot be written out as device source (the host callback path still runs it)"""


_FUNCS = ("exp", "log", "sqrt", "sin", "cos", "tan", "tanh", "sinh", "cosh", "arcsin", "arccos", "arctan", "log1p", "expm1", "log10",
          "log2", "exp2", "cbrt", "floor", "ceil", "erf", "erfc", "rint", "trunc")
_CNAME = {"arcsin": "asin", "arccos": "acos", "arctan": "atan"}
_NPFN = {"erf": math.erf, "erfc": math.erfc}


class _Trace:
    """the expression DAG of one trace: nodes are interned, so common subexpressions are one temporary"""

    def __init__(self):
        self.nodes = []
        self.index = {}
        self.params = []      # values of the captured parameters: node ("ud", k) stands for params[k]
        # Python branches on sampled values (`1.0 if x[0] ** 2 + x[1] ** 2 < 1 else 0.0`, `if`, `and` / `or`, `while`): the closure is run
        # once per WAY through its branches.  `script` forces the outcome of the k-th truth test of a run, `conds` records what was
        # tested; explore() below enumerates the ways and joins their results with selects.
        self.script, self.conds, self.decided = [], [], []
        self.tables = []      # float arrays the closure indexes with a sampled value (_Table): node ("table", id, j, stride, rows, index)
        self.dynamic = []     # trace_measure: `obs[i][sampled index] += value` met by the current run: (observable, index, value)

    def decide(self, cond):
        k = len(self.decided)
        out = self.script[k] if k < len(self.script) else True
        self.decided.append(out)
        self.conds.append(cond)
        return out

    def param(self, v):
        self.params.append(float(v))
        return self.node("ud", len(self.params) - 1)

    def node(self, op, *args):
        key = (op,) + tuple(a.id if isinstance(a, Sym) else ("k", repr(a)) for a in args)   # (repr: 0.0 and -0.0 are two constants)
        s = self.index.get(key)
        if s is None:
            s = Sym(self, op, args, len(self.nodes))
            self.nodes.append(s)
            self.index[key] = s
        return s

    def const(self, v):
        if isinstance(v, (bool, np.bool_)):
            v = float(v)
        if isinstance(v, (int, float, np.integer, np.floating)):
            v = float(v)
            if not math.isfinite(v):
                raise TraceError("non-finite constant %r" % v)
            return self.node("const", v)
        raise TraceError("cannot use %r (%s) in an integrand expression" % (v, type(v).__name__))

    def lift(self, v):
        return v if isinstance(v, Sym) else self.const(v)


class Sym:
    """one value of the traced computation (a draw, a constant, or an operation on earlier values)"""
    __slots__ = ("t", "op", "args", "id")

    def __init__(self, t, op, args, id_):
        self.t, self.op, self.args, self.id = t, op, args, id_

    # -- what must not happen during a trace
    def __bool__(self):
        # a Python branch on a sampled value: this run takes the way the script says (explore() runs the others)
        return self.t.decide(self if self.op in _BOOL else self.t.node("!=", self, self.t.const(0.0)))

    def __float__(self):
        raise TraceError("float() of a sampled value (math.* functions: use the numpy ones)")

    __int__ = __complex__ = __float__

    def __index__(self):
        raise TraceError("a Python list or a plain numpy array indexed with a sampled value (float arrays reached through "
                         "config.userdata or captured by the closure are indexed symbolically: trace._Table)")
    __hash__ = object.__hash__

    # -- arithmetic (an ndarray operand hands the operation back to numpy, which applies it element by element)
    def _bin(self, op, other, swap=False):
        if isinstance(other, (np.ndarray, CSym)):
            return NotImplemented
        if isinstance(other, (complex, np.complexfloating)):
            if op not in "+-*/":
                raise TraceError("a comparison with a complex number")
            return CSym(self, 0.0)._bin(op, other, swap)
        o = self.t.lift(other)
        a, b = (o, self) if swap else (self, o)
        if op == "+" and (_is_const(a, 0.0) or _is_const(b, 0.0)):
            return b if _is_const(a, 0.0) else a
        if op == "*" and (_is_const(a, 1.0) or _is_const(b, 1.0)):
            return b if _is_const(a, 1.0) else a
        if op == "-" and _is_const(b, 0.0):
            return a
        if op == "/" and _is_const(b, 1.0):
            return a
        return self.t.node(op, a, b)

    def __add__(self, o): return self._bin("+", o)
    def __radd__(self, o): return self._bin("+", o, True)
    def __sub__(self, o): return self._bin("-", o)
    def __rsub__(self, o): return self._bin("-", o, True)
    def __mul__(self, o): return self._bin("*", o)
    def __rmul__(self, o): return self._bin("*", o, True)
    def __truediv__(self, o): return self._bin("/", o)
    def __rtruediv__(self, o): return self._bin("/", o, True)
    def __lt__(self, o): return self._bin("<", o)
    def __le__(self, o): return self._bin("<=", o)
    def __gt__(self, o): return self._bin(">", o)
    def __ge__(self, o): return self._bin(">=", o)
    def __eq__(self, o): return self._bin("==", o)      # (a Discrete draw against a number; nodes are interned by id, not compared)
    def __ne__(self, o): return self._bin("!=", o)
    def __neg__(self): return self.t.node("neg", self)
    def __pos__(self): return self
    def __abs__(self): return self.t.node("fabs", self)

    def __pow__(self, p):
        if isinstance(p, np.ndarray):
            return NotImplemented
        if isinstance(p, (int, float, np.integer, np.floating)) and float(p) == int(p) and 0 <= int(p) <= 4:
            k = int(p)           # small integer powers as products, like Julia's x^2 (and what a hand-written body would say)
            if k == 0:
                return self.t.const(1.0)
            r = self
            for _ in range(k - 1):
                r = r * self
            return r
        if isinstance(p, (int, float, np.integer, np.floating)) and float(p) == 0.5:
            return self.t.node("sqrt", self)
        if isinstance(p, (int, float, np.integer, np.floating)) and float(p) == -1.0:
            return 1.0 / self
        return self.t.node("pow", self, self.t.lift(p))

    def __rpow__(self, b):
        if isinstance(b, np.ndarray):
            return NotImplemented
        return self.t.node("pow", self.t.lift(b), self)

    # Python's (and Julia's `mod` / `fld`) floored division: the remainder has the sign of the divisor.  C's fmod has the sign of the
    # dividend, so the written-out body corrects it where the two differ
    def __mod__(self, o):
        if isinstance(o, np.ndarray):
            return NotImplemented
        return _pymod(self, o)

    def __rmod__(self, o):
        if isinstance(o, np.ndarray):
            return NotImplemented
        return _pymod(o, self)

    def __floordiv__(self, o):
        if isinstance(o, np.ndarray):
            return NotImplemented
        return (self / o).floor()

    def __rfloordiv__(self, o):
        if isinstance(o, np.ndarray):
            return NotImplemented
        return (o / self).floor()

    def __round__(self, ndigits=None):
        if ndigits not in (None, 0):
            raise TraceError("round(x, ndigits) of a sampled value")
        return self.rint()                                 # (half to even, like Python's round and numpy's rint)

    def round(self, decimals=0, out=None):                 # np.round(x) on an object calls x.round()
        return self.__round__(decimals)

    # -- numpy on a symbol: np.maximum(x[0], 0.5), np.where(x[0] > 0.5, a, b), np.clip(...), and ndarray <op> symbol
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs:
            raise TraceError("np.%s.%s with %s on a sampled value" % (ufunc.__name__, method, sorted(kwargs) or "this call form"))
        if any(isinstance(i, np.ndarray) for i in inputs):   # (symbols boxed: frompyfunc is a ufunc itself and would come back here)
            return np.frompyfunc(lambda *a: _ufunc(ufunc, *a), len(inputs), 1)(*[_boxed(i) for i in inputs])
        return _ufunc(ufunc, *inputs)

    def __array_function__(self, func, types, args, kwargs):
        if func is np.where and len(args) == 3 and not kwargs:
            return where(*args)
        if func is np.clip and len(args) == 3 and not kwargs:
            return fmin(fmax(args[0], args[1]), args[2])
        return func._implementation(*args, **kwargs)

    # -- what a batch-vectorised closure does with a vector over the batch, on the one sample of the trace
    def sum(self, *a, **k): return self          # weights[0].sum(): the per-sample value IS the sample's contribution

    def __getitem__(self, m):                    # weights[0][mask]: the value where the mask holds, else nothing (0)
        if isinstance(m, Sym):
            if m.op in _BOOL:
                return where(m, self, 0.0)
        elif m is Ellipsis or (isinstance(m, slice) and m == slice(None)):
            return self
        raise TraceError("indexing a sampled value with %r" % (m,))

    def __invert__(self):
        if self.op not in _BOOL:
            raise TraceError("~ of a value that is not a comparison")
        return self.t.node("not", self)

    def _logic(self, op, o):
        if isinstance(o, (bool, np.bool_)):
            return (self if o else self.t.node("not", self.t.node("or", self, self.t.node("not", self)))) if op == "and" else (self if not o else self.t.node("or", self, self.t.node("not", self)))
        if not (isinstance(o, Sym) and o.op in _BOOL and self.op in _BOOL):
            raise TraceError("& / | of values that are not comparisons")
        return self.t.node(op, self, o)

    def __and__(self, o): return self._logic("and", o)
    def __rand__(self, o): return self._logic("and", o)
    def __or__(self, o): return self._logic("or", o)
    def __ror__(self, o): return self._logic("or", o)

    def conjugate(self): return self
    conj = conjugate

    @property
    def real(self): return self

    @property
    def imag(self): return 0.0


_BOOL = ("<", "<=", ">", ">=", "==", "!=", "not", "and", "or")


def _z(v):
    """a component that is structurally zero (a real value promoted to complex): its terms are not written out"""
    return not isinstance(v, Sym) and v == 0.0


def _rmul(a, b):
    return 0.0 if _z(a) or _z(b) else a * b


def _radd(a, b, sign=1.0):
    if _z(b):
        return a
    if _z(a):
        return b if sign > 0 else -b
    return a + b if sign > 0 else a - b


class CSym:
    """a complex value of the traced computation (type = ComplexF64 configurations, main.jl:279,284): a pair of real values -- draws,
    constants or operations on them -- so that the written-out body stays real arithmetic on the (re, im) slots of `w`.  A real value
    that meets a complex one is (value, structural zero), like Julia's Real * Complex: no cross terms with the missing part."""
    __slots__ = ("re", "im")

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        n = ufunc.__name__
        if method != "__call__" or kwargs or any(isinstance(i, np.ndarray) for i in inputs):
            raise TraceError("np.%s.%s on a complex sampled value in this call form" % (n, method))
        if n in _UFUNC_BIN and _UFUNC_BIN[n] in "+-*/":
            return CSym.of(inputs[0])._bin(_UFUNC_BIN[n], inputs[1])
        one = {"exp": CSym.exp, "conjugate": CSym.conjugate, "negative": CSym.__neg__, "positive": CSym.__pos__, "absolute": CSym.__abs__,
               "square": lambda z: z * z, "reciprocal": lambda z: 1.0 / z}
        if n in one and len(inputs) == 1:
            return one[n](self)
        if n == "power" and inputs[0] is self:
            return self ** inputs[1]
        raise TraceError("np.%s of a complex value is not written out" % n)

    def __init__(self, re, im):
        self.re, self.im = re, im

    @staticmethod
    def of(v):
        if isinstance(v, CSym):
            return v
        if isinstance(v, (complex, np.complexfloating)):
            return CSym(float(v.real), float(v.imag))
        if isinstance(v, (Sym, int, float, np.integer, np.floating)):
            return CSym(v if isinstance(v, Sym) else float(v), 0.0)
        raise TraceError("cannot use %r (%s) in a complex integrand expression" % (v, type(v).__name__))

    def _bin(self, op, other, swap=False):
        if isinstance(other, np.ndarray):
            return NotImplemented
        a, b = (CSym.of(other), self) if swap else (self, CSym.of(other))
        if op == "+":
            return CSym(_radd(a.re, b.re), _radd(a.im, b.im))
        if op == "-":
            return CSym(_radd(a.re, b.re, -1.0), _radd(a.im, b.im, -1.0))
        if op == "*":
            return CSym(_radd(_rmul(a.re, b.re), _rmul(a.im, b.im), -1.0), _radd(_rmul(a.re, b.im), _rmul(a.im, b.re)))
        if op == "/":
            if _z(b.im):
                return CSym(a.re / b.re, 0.0 if _z(a.im) else a.im / b.re)
            d = _radd(_rmul(b.re, b.re), _rmul(b.im, b.im))
            n = a._bin("*", CSym(b.re, 0.0 if _z(b.im) else -b.im))
            return CSym(0.0 if _z(n.re) else n.re / d, 0.0 if _z(n.im) else n.im / d)
        raise TraceError("complex values are not ordered")

    def __add__(self, o): return self._bin("+", o)
    def __radd__(self, o): return self._bin("+", o, True)
    def __sub__(self, o): return self._bin("-", o)
    def __rsub__(self, o): return self._bin("-", o, True)
    def __mul__(self, o): return self._bin("*", o)
    def __rmul__(self, o): return self._bin("*", o, True)
    def __truediv__(self, o): return self._bin("/", o)
    def __rtruediv__(self, o): return self._bin("/", o, True)
    def __neg__(self): return CSym(-self.re if not _z(self.re) else 0.0, -self.im if not _z(self.im) else 0.0)
    def __pos__(self): return self

    def __abs__(self):
        m = _radd(_rmul(self.re, self.re), _rmul(self.im, self.im))
        return m.sqrt() if isinstance(m, Sym) else math.sqrt(m)

    def __pow__(self, p):
        if isinstance(p, (int, np.integer)) and 0 <= int(p) <= 8:
            r = CSym(1.0, 0.0)
            for _ in range(int(p)):
                r = r * self
            return r
        raise TraceError("a complex value to the power %r" % (p,))

    def __lt__(self, o): raise TraceError("complex values are not ordered")
    __le__ = __gt__ = __ge__ = __lt__

    def __bool__(self):
        raise TraceError("a Python branch on a sampled value (use mci.trace.where(cond, a, b))")

    def conjugate(self): return CSym(self.re, 0.0 if _z(self.im) else -self.im)
    conj = conjugate

    # -- what a batch-vectorised closure does with a vector over the batch, on the one sample of the trace (like Sym)
    def sum(self, *a, **k): return self

    def __getitem__(self, m):
        if isinstance(m, Sym) and m.op in _BOOL:
            return where(m, self, 0.0)
        if m is Ellipsis or (isinstance(m, slice) and m == slice(None)):
            return self
        raise TraceError("indexing a sampled value with %r" % (m,))

    def exp(self):
        e = self.re.exp() if isinstance(self.re, Sym) else math.exp(self.re)
        if _z(self.im):
            return CSym(e, 0.0)
        c, s_ = (self.im.cos(), self.im.sin()) if isinstance(self.im, Sym) else (math.cos(self.im), math.sin(self.im))
        return CSym(e * c, e * s_)

    @property
    def real(self): return self.re

    @property
    def imag(self): return self.im


def _is_const(s, v):
    return s.op == "const" and s.args[0] == v and math.copysign(1.0, s.args[0]) == math.copysign(1.0, v)


def _method(name):
    def f(self):
        return self.t.node(name, self)
    f.__name__ = name
    return f


for _n in _FUNCS:
    setattr(Sym, _n, _method(_n))   # np.exp(obj) on an object calls obj.exp()
Sym.fabs = Sym.__abs__
Sym.absolute = Sym.__abs__
Sym.square = lambda self: self * self
Sym.reciprocal = lambda self: 1.0 / self
Sym.negative = Sym.__neg__


def _boxed(v):
    if not isinstance(v, Sym):
        return v
    a = np.empty((), dtype=object)
    a[()] = v
    return a


_UFUNC_BIN = {"add": "+", "subtract": "-", "multiply": "*", "divide": "/", "true_divide": "/", "less": "<", "less_equal": "<=",
              "greater": ">", "greater_equal": ">=", "equal": "==", "not_equal": "!="}


def _ufunc(ufunc, *a):
    """one numpy ufunc applied to symbols / numbers (Sym.__array_ufunc__)"""
    if not any(isinstance(v, Sym) for v in a):
        return ufunc(*a)
    n = ufunc.__name__
    t = [v for v in a if isinstance(v, Sym)][0].t
    if n in _UFUNC_BIN:
        x, y = t.lift(a[0]), t.lift(a[1])
        return x._bin(_UFUNC_BIN[n], y)
    if n == "power":
        return t.lift(a[0]) ** a[1] if isinstance(a[0], Sym) else a[1].__rpow__(a[0])
    if n in ("maximum", "fmax"):
        return fmax(a[0], a[1])
    if n in ("minimum", "fmin"):
        return fmin(a[0], a[1])
    if n == "arctan2":
        return arctan2(a[0], a[1])
    if n == "hypot":
        x, y = t.lift(a[0]), t.lift(a[1])
        return (x * x + y * y).sqrt()
    if n in ("remainder", "mod"):
        return _pymod(a[0], a[1])
    if n == "fmod":
        x, y = t.lift(a[0]), t.lift(a[1])
        return t.node("fmod", x, y)
    if n == "floor_divide":
        return (t.lift(a[0]) / a[1]).floor()
    if n == "negative":
        return -a[0]
    if n == "positive":
        return a[0]
    if n in ("absolute", "fabs"):
        return abs(a[0])
    if n == "sign":
        return where(a[0] > 0.0, 1.0, where(a[0] < 0.0, -1.0, 0.0))
    if n == "heaviside":
        return where(t.lift(a[0]) > 0.0, 1.0, where(t.lift(a[0]) < 0.0, 0.0, a[1]))
    if len(a) == 1 and hasattr(Sym, n):
        return getattr(a[0], n)()
    raise TraceError("np.%s is not written out" % n)


def _pymod(a, b):
    """a % b with Python's sign convention, of traced values"""
    t, a, b = _lift2(a, b)
    r = t.node("fmod", a, b)
    if b.op == "const":                                     # (the usual case, x % 1.0: one select)
        fix = (r < 0.0) if b.args[0] > 0.0 else (r > 0.0)
    else:
        fix = (r * b) < 0.0
    return where(fix, r + b, r)


def _lift2(a, b):
    t = a.t if isinstance(a, Sym) else b.t if isinstance(b, Sym) else None
    if t is None:
        raise TraceError("no sampled value among the arguments")
    return t, t.lift(a), t.lift(b)


def where(cond, a, b):
    """cond ? a : b, element by element; cond a comparison of traced values (the traced counterpart of np.where)"""
    if any(isinstance(v, np.ndarray) for v in (cond, a, b)):
        return np.frompyfunc(where, 3, 1)(_boxed(cond), _boxed(a), _boxed(b))
    if not isinstance(cond, Sym):
        return a if cond else b
    if any(isinstance(v, (CSym, complex, np.complexfloating)) for v in (a, b)):
        a, b = CSym.of(a), CSym.of(b)
        return CSym(where(cond, a.re, b.re), where(cond, a.im, b.im))
    t = cond.t
    return t.node("where", cond, t.lift(a), t.lift(b))


def fmax(a, b):
    """the larger of two traced values (np.maximum compares Python objects and cannot be traced)"""
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.frompyfunc(fmax, 2, 1)(_boxed(a), _boxed(b))
    if not isinstance(a, Sym) and not isinstance(b, Sym):
        return max(a, b)
    t, a, b = _lift2(a, b)
    return t.node("fmax", a, b)


def fmin(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.frompyfunc(fmin, 2, 1)(_boxed(a), _boxed(b))
    if not isinstance(a, Sym) and not isinstance(b, Sym):
        return min(a, b)
    t, a, b = _lift2(a, b)
    return t.node("fmin", a, b)


def arctan2(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.frompyfunc(arctan2, 2, 1)(_boxed(a), _boxed(b))
    if not isinstance(a, Sym) and not isinstance(b, Sym):
        return math.atan2(a, b)
    t, a, b = _lift2(a, b)
    return t.node("atan2", a, b)


# ---- writing the DAG out, and evaluating it for the check ----
_BINOPS = ("+", "-", "*", "/", "<", "<=", ">", ">=", "==", "!=")
# The body grammar: how an operation is written as a C expression of its operands ({k} = operand k), and which operands are truth
# values ("c": written `t` for a comparison, `(v != 0.0)` for a number) instead of numbers ("n": a comparison is cast, `(double)t`).
# Everything else is a call `name(operands)` of the C math function of that name (_CNAME renames numpy's arc* functions).
# julia/MCIntegrationHIP.jl carries the same two tables for its tracer (tests/test_binding_layouts.py compares them).
C_FORMAT = {"+": "{0} + {1}", "-": "{0} - {1}", "*": "{0} * {1}", "/": "{0} / {1}", "<": "{0} < {1}", "<=": "{0} <= {1}", ">": "{0} > {1}",
            ">=": "{0} >= {1}", "==": "{0} == {1}", "!=": "{0} != {1}", "neg": "-{0}", "not": "!{0}", "and": "{0} && {1}", "or": "{0} || {1}",
            "where": "{0} ? {1} : {2}"}
C_OPERANDS = {"not": "c", "and": "cc", "or": "cc", "where": "cnn"}


def _reachable(outs, stop=()):
    """nodes the outputs depend on, children before parents; nothing below the nodes in `stop` (they are leaves to the caller)"""
    seen, order = set(), []

    def visit(s):
        stack = [(s, False)]
        while stack:
            n, done = stack.pop()
            if done:
                order.append(n)
                continue
            if n.id in seen:
                continue
            seen.add(n.id)
            stack.append((n, True))
            if n.id in stop:
                continue
            for a in n.args:
                if isinstance(a, Sym):
                    stack.append((a, False))
    for o in outs:
        visit(o)
    return order


def _literal(v):
    r = repr(float(v))
    return r if any(c in r for c in ".en") else r + ".0"


class _Table(np.ndarray):
    """A float array of the user's -- config.userdata, an attribute of it, an array the closure captured -- during a trace: it is the
    array it was for everything Python does with it, and indexing it with a SAMPLED value (`grid[bin[0] - 1]`, example of
    docs/src/index.md "Measure Histogram"; `para.extQ[ext[0] - 1]`, test/bubble.jl:60) is a table lookup of the written-out body:
    the values go into the userdata vector and the element is `ud[base + (int)index]` (a row of an N-d table: one lookup per element).
    Sampled values on several leading axes (`vertex[a[0] - 1, b[0] - 1]`) become one row-major flat index.
    0-based like every index here; an index outside the table is clamped (the check against the closure at random points refuses a
    closure that relies on anything else, e.g. Python's negative indices)."""

    def __new__(cls, content, t, values=None):
        obj = np.asarray(content).view(cls)
        obj._t, obj._values, obj._tid = t, np.ascontiguousarray(content if values is None else values, dtype=np.float64), None
        return obj

    def __array_finalize__(self, obj):
        self._t = self._values = self._tid = None          # (a slice or a copy is a plain array again)

    def __getitem__(self, i):
        if isinstance(i, Sym):
            return self._lookup(i)
        if isinstance(i, tuple) and any(isinstance(q, Sym) for q in i):
            last = max(k for k, q in enumerate(i) if isinstance(q, Sym))
            lead, rest = i[:last + 1], i[last + 1:]
            if not all(isinstance(q, (Sym, int, np.integer)) for q in lead):
                raise TraceError("a table indexed with a sampled value behind a slice")
            out = self._lookup(lead[0]) if len(lead) == 1 else self._lookup_axes(lead)
            return out[rest] if rest else out
        out = np.ndarray.__getitem__(self, i)
        return out.view(np.ndarray) if isinstance(out, np.ndarray) else out

    def _lookup_axes(self, lead):
        """`tab[a, b]` with sampled values on several leading axes (a vertex table over two Discrete draws): the row-major flat
        index of the clamped entries, then one lookup in the table seen as [prod of those axes, rest]"""
        if self._values is None or self._values.ndim < len(lead):
            raise TraceError("more indices than the table has axes")
        shape = self._values.shape
        flat = 0.0
        for axis, q in enumerate(lead):
            n = shape[axis]
            if n == 0:
                raise TraceError("an empty table indexed with a sampled value")
            term = fmin(fmax(q, 0.0), float(n - 1)) if isinstance(q, Sym) else float(int(q) % n)
            flat = flat * float(n) + term
        rows = int(np.prod(shape[:len(lead)]))
        view = _Table(self._values.reshape((rows,) + shape[len(lead):]), self._t)
        view._tid = self._tid if self._tid is not None else None
        if self._tid is None:                               # (one block of ud[] for the table, however it is indexed)
            self._tid = view._tid = len(self._t.tables)
            self._t.tables.append(self._values)
        return view._lookup(flat)

    def _lookup(self, idx):
        if self._t is None or self._values is None or self._values.ndim == 0:
            raise TraceError("a slice or copy of a table indexed with a sampled value (index the array itself)")
        t = self._t
        if idx.t is not t:
            raise TraceError("a value of another trace")
        if self._tid is None:
            self._tid = len(t.tables)
            t.tables.append(self._values)
        rows = self._values.shape[0]
        stride = int(self._values.size // rows) if rows else 0
        if rows == 0:
            raise TraceError("an empty table indexed with a sampled value")
        if self._values.ndim == 1:
            return t.node("table", self._tid, 0, 1, rows, idx)
        out = np.empty(self._values.shape[1:], dtype=object)
        for j, q in enumerate(np.ndindex(out.shape)):
            out[q] = t.node("table", self._tid, j, stride, rows, idx)
        return out


_TABLE_MAX = 1 << 16


def _as_table(v, t):
    """v as a _Table if it is a rectangular numeric array (or a list / tuple of numbers or of such arrays) of a sensible size, else None"""
    if isinstance(v, _Table):
        return v
    if isinstance(v, np.ndarray):
        return _Table(v, t) if v.dtype.kind in "fiu" and 0 < v.size <= _TABLE_MAX and v.ndim >= 1 else None
    if isinstance(v, (list, tuple)) and v and all(isinstance(q, (int, float, np.integer, np.floating, np.ndarray, list, tuple)) and not isinstance(q, bool) for q in v):
        try:
            a = np.asarray(v, dtype=np.float64)
        except (ValueError, TypeError):
            return None
        return _Table(a, t) if 0 < a.size <= _TABLE_MAX and a.ndim >= 1 and np.all(np.isfinite(a)) else None
    return None


class _UserdataView:
    """config.userdata (a struct of parameters: test/bubble.jl:12-27 `para`) during a trace: attribute reads hand arrays out as _Tables
    and floats as PARAMETERS of the trace (module docstring: one body for every value of beta, kF, ...)"""

    def __init__(self, obj, t, floats):
        object.__setattr__(self, "_obj", obj)
        object.__setattr__(self, "_t", t)
        object.__setattr__(self, "_floats", floats)
        object.__setattr__(self, "_seen", {})

    def _view(self, key, get):
        seen = object.__getattribute__(self, "_seen")
        if key not in seen:
            seen[key] = _userdata_view(get(object.__getattribute__(self, "_obj")), object.__getattribute__(self, "_t"),
                                       object.__getattribute__(self, "_floats"))
        return seen[key]

    def __getattr__(self, name):
        return self._view(name, lambda o: getattr(o, name))

    def __getitem__(self, k):
        return self._view(("item", k), lambda o: o[k])

    def __setattr__(self, name, v):
        raise TraceError("the closure writes to config.userdata (hidden state)")


def _param_table(v, t):
    """a float array of up to 64 elements as parameters of the trace (one ud slot each) that is also a table for a sampled index"""
    v = np.asarray(v, dtype=np.float64)
    a = np.empty(v.shape, dtype=object)
    for i in np.ndindex(v.shape):
        a[i] = t.param(v[i])
    return _Table(a, t, values=v)


def _userdata_view(v, t, floats=False):
    if floats and isinstance(v, (float, np.floating)) and not isinstance(v, bool) and math.isfinite(v):
        return t.param(v)
    tb = _as_table(v, t)
    if tb is not None:
        if floats and tb._values.size <= 64 and (not isinstance(v, np.ndarray) or v.dtype.kind == "f") and np.all(np.isfinite(tb._values)):
            return _param_table(tb._values, t)
        return tb
    if isinstance(v, (str, bytes, int, float, complex, bool, type(None), np.generic, np.ndarray, list, tuple, types.FunctionType,
                      types.BuiltinFunctionType, types.MethodType, types.ModuleType, type)):
        return v
    if isinstance(v, dict) or hasattr(v, "__dict__") or hasattr(v, "__slots__"):
        return _UserdataView(v, t, floats)
    return v


def _trace_config(config, t, floats=False):
    """the Configuration a traced closure is called with: the user's, with `userdata` seen through _userdata_view"""
    import copy
    ud = getattr(config, "userdata", None)
    view = _userdata_view(ud, t, floats)
    if view is ud:
        return config
    c = copy.copy(config)
    c.userdata = view
    return c


def hoist(outs, params):
    """Every maximal subexpression that depends on captured parameters (and constants) only -> one userdata slot, evaluated here on
    the host: ({node id: "ud[j]"}, [values]).  The body then does not change with the parameters' values."""
    if not params:
        return {}, []
    order = _reachable(outs)
    on_x, on_p = {}, {}
    for n in order:
        kids = [a for a in n.args if isinstance(a, Sym)]
        on_x[n.id] = n.op in ("x", "rw") or any(on_x[a.id] for a in kids)
        on_p[n.id] = n.op == "ud" or any(on_p[a.id] for a in kids)
    slots, values, byid = {}, [], {n.id: n for n in order}

    def take(n):
        if n.id not in slots:
            slots[n.id] = "ud[%d]" % len(values)
            values.append(float(evaluate([n], np.zeros((0, 1)), params=params)[0][0]))
    outset = {o.id for o in outs}
    for n in order:
        if on_x[n.id]:
            for a in n.args:
                if isinstance(a, Sym) and on_p[a.id] and not on_x[a.id]:
                    take(a)
        elif on_p[n.id] and n.id in outset:
            take(n)
    for v in values:
        if not math.isfinite(v):
            raise TraceError("a captured parameter evaluates to a non-finite value")
    return slots, values


def emit(outs, sink=lambda i, ref: "w[%d] = %s;" % (i, ref), leaves=None, tables=None):
    """HIP C++ / C body: one `const double tK = ...;` per operation (children first), then `w[i] = ...;` (or what `sink` says).
    `leaves` ({node id: text}, from hoist()): nodes written as that text and not looked into.  A comparison used as a NUMBER
    ((x > a) * 2.0, (x > a) + (y > b)) is cast to double where it is used: the arithmetic is floating point like the closure's."""
    leaves = leaves or {}
    order = _reachable(outs, stop=leaves)
    name, lines = {}, []

    def ref(a):
        return name[a.id]

    def num(a):   # as an operand of arithmetic
        return "(double)%s" % name[a.id] if a.op in _BOOL and a.id not in leaves else name[a.id]

    def cond(a):  # as a truth value
        return name[a.id] if a.op in _BOOL and a.id not in leaves else "(%s != 0.0)" % name[a.id]
    for n in order:
        if n.id in leaves:
            name[n.id] = leaves[n.id]
            continue
        if n.op in ("x", "rw", "ud"):
            name[n.id] = "%s[%d]" % (n.op, n.args[0])
            continue
        if n.op == "const":
            v = n.args[0]
            name[n.id] = _literal(v) if math.copysign(1.0, v) > 0 else "(%s)" % _literal(v)
            continue
        if n.op == "table":      # (table id, element of the row, row stride, rows, index): tables[id] = where the table starts in ud[]
            tid, j, stride, rows, idx = n.args
            at = "(int)fmin(fmax(%s, 0.0), %d.0)" % (num(idx), rows - 1)
            lines.append("const double t%d = ud[%d + %s];" % (n.id, tables[tid] + j, at if stride == 1 else "%d * %s" % (stride, at)))
            name[n.id] = "t%d" % n.id
            continue
        kinds = C_OPERANDS.get(n.op, "n" * len(n.args))
        ops = [cond(a) if k == "c" else num(a) for a, k in zip(n.args, kinds)]
        if n.op in C_FORMAT:
            e = C_FORMAT[n.op].format(*ops)
        else:
            e = "%s(%s)" % (_CNAME.get(n.op, n.op), ", ".join(ops))
        if n.op in _BOOL:
            lines.append("const int t%d = %s;" % (n.id, e))      # (int: the body is also compiled as C by the oracle)
        else:
            lines.append("const double t%d = %s;" % (n.id, e))
        name[n.id] = "t%d" % n.id
    for i, o in enumerate(outs):
        lines.append(sink(i, num(o)))
    return "\n".join(lines)


def evaluate(outs, X, R=None, params=(), tables=()):
    """The DAG on numeric draws X[draw, sample] (relative weights R[integrand, sample], captured parameters `params`) WITH THE
    SEMANTICS OF THE EMITTED C, for the check against the closure itself: a comparison is the number 0.0 or 1.0 (where numpy's
    bool + bool is a logical or and C's int + int is 2), a truth value is `!= 0`.  A closure whose numpy arithmetic on booleans
    means something else than the written-out body computes therefore fails the check and keeps the host path."""
    val = {}
    f64 = np.float64

    def truth(v):
        return np.asarray(v) != 0.0
    with np.errstate(all="ignore"):
        for n in _reachable(outs):
            a = [val[q.id] if isinstance(q, Sym) else q for q in n.args]
            if n.op == "x":
                v = X[n.args[0]]
            elif n.op == "rw":
                v = R[n.args[0]]
            elif n.op == "ud":
                v = f64(params[n.args[0]])
            elif n.op == "table":
                tid, j, stride, rows = n.args[:4]
                at = np.clip(np.nan_to_num(np.asarray(a[4], dtype=f64)), 0.0, rows - 1.0).astype(np.int64)    # ((int) truncates; the index is an integer)
                v = tables[tid].reshape(-1)[j + stride * at]
            elif n.op == "not":
                v = np.logical_not(truth(a[0])).astype(f64)
            elif n.op in ("and", "or"):
                v = (np.logical_and if n.op == "and" else np.logical_or)(truth(a[0]), truth(a[1])).astype(f64)
            elif n.op == "const":
                v = f64(n.args[0])
            elif n.op == "+":
                v = a[0] + a[1]
            elif n.op == "-":
                v = a[0] - a[1]
            elif n.op == "*":
                v = a[0] * a[1]
            elif n.op == "/":
                v = np.asarray(a[0], dtype=f64) / np.asarray(a[1], dtype=f64)
            elif n.op in ("<", "<=", ">", ">=", "==", "!="):
                v = {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal, "==": np.equal, "!=": np.not_equal}[n.op](a[0], a[1]).astype(f64)
            elif n.op == "neg":
                v = -a[0]
            elif n.op == "where":
                v = np.where(truth(a[0]), a[1], a[2])
            elif n.op == "pow":
                v = np.power(np.asarray(a[0], dtype=f64), a[1])
            elif n.op == "atan2":
                v = np.arctan2(a[0], a[1])
            elif n.op == "fmod":
                v = np.fmod(np.asarray(a[0], dtype=f64), a[1])
            elif n.op in ("fmax", "fmin", "fabs"):
                v = getattr(np, n.op)(*a)
            elif n.op in _NPFN:
                v = np.vectorize(_NPFN[n.op])(a[0])
            else:
                v = getattr(np, n.op)(a[0])
            val[n.id] = v
    return [np.broadcast_to(np.asarray(val[o.id], dtype=np.float64), X.shape[1:]) for o in outs]


def _parametrized(fn, t, floats=True, _memo=None, _depth=0):
    """A copy of the closure whose captured floats are parameters of trace `t` and whose captured arrays are tables (module
    docstring); the closure itself if it has neither or is not a plain Python function.  The helper functions it reaches -- Python
    functions in its cells, functions of its own module that its code names (`green(tau, omega, beta)` next to the integrand,
    test/bubble.jl:40-51) -- are copied the same way (up to 16 levels of helpers calling helpers)."""
    if not isinstance(fn, types.FunctionType):
        return fn
    memo = {} if _memo is None else _memo
    if id(fn) in memo:
        return memo[id(fn)]
    memo[id(fn)] = fn                                      # (a function that reaches itself keeps the original there)

    def conv(v):
        if floats and isinstance(v, (float, np.floating)) and not isinstance(v, bool) and math.isfinite(v):
            return t.param(v)
        if floats and isinstance(v, np.ndarray) and not isinstance(v, _Table) and v.dtype.kind == "f" and 0 < v.size <= 64 and np.all(np.isfinite(v)):
            changed[0] = True
            return _param_table(v, t)                      # (its elements are parameters; indexed with a sampled value it is a table)
        if isinstance(v, np.ndarray) and not isinstance(v, _Table):
            tb = _as_table(v, t)
            if tb is not None:
                changed[0] = True
                return tb
        if isinstance(v, types.FunctionType) and _depth < 16:
            new = _parametrized(v, t, floats, memo, _depth + 1)
            if new is not v:
                changed[0] = True
            return new
        return v
    changed = [False]
    n0 = len(t.params)
    cells = None
    if fn.__closure__:
        cells = []
        for c in fn.__closure__:
            try:
                cells.append(types.CellType(conv(c.cell_contents)))
            except ValueError:            # an empty cell
                cells.append(c)
        cells = tuple(cells)
    g = fn.__globals__
    names = [n for n in fn.__code__.co_names if n in g and not isinstance(g[n], bool) and
             (isinstance(g[n], (float, np.floating, np.ndarray)) or
              (isinstance(g[n], types.FunctionType) and g[n].__module__ == fn.__module__ and g[n] is not fn))]
    if names:
        g = dict(g)
        for n in names:
            g[n] = conv(g[n])
    defaults = tuple(conv(d) for d in fn.__defaults__) if fn.__defaults__ else None
    if len(t.params) == n0 and not changed[0]:
        return fn
    new = types.FunctionType(fn.__code__, g, fn.__name__, defaults, cells)
    new.__kwdefaults__ = fn.__kwdefaults__
    memo[id(fn)] = new
    return new


def _pools(config):
    """(first flat draw, maxdof, entries per slot, offset, is a CompositeVar, kinds) per variable type (Configuration.pool_layout), and
    the number of draws"""
    pools = config.pool_layout()
    return pools, sum(p[1] * p[2] for p in pools)


def _argument(pools, leaf, pad=lambda: 0.0, numeric=False):
    """what the closure is called with: leaf(k) for flat draw k, arranged like HostIntegrand's argument without the batch axis.  A
    CompositeVar is indexed like the reference's (variable.jl:436-447: `cvar[i]` is its i-th leaf VARIABLE, itself indexed by slot, and
    iterating it gives the leaves -- `x, y = cvar`), i.e. [leaf][slot]; a FermiK pool is [slot][component] (K[i] is a momentum).  A pool
    with `offset` has that many leading slots nobody samples -- the reference's closures address X[i + offset] (variable.jl:577,
    test/montecarlo.jl:19-32) -- filled with pad() (asked for only then: a trace without offsets numbers its nodes as it always did).
    numeric: leaf(k) are numbers (the check of a trace against the closure itself) -- plain float arrays, and INTEGER ones for Discrete
    draws like the reference's Discrete pool (variable.jl:283) and like Engine._pool_views hands them to a host closure."""
    def typed(a, kind):
        if not numeric:
            return a
        return np.rint(a.astype(np.float64)).astype(np.int64) if kind == "d" else a.astype(np.float64)

    def arr(k0, md, nl, off=0, composite=False, kinds="c"):
        a = np.empty((off + md,) if nl == 1 and not composite else (off + md, nl), dtype=object)
        if off:
            a[:off] = pad()
        for s in range(md):
            if a.ndim == 1:
                a[off + s] = leaf(k0 + s)
            else:
                for l in range(nl):
                    a[off + s, l] = leaf(k0 + s * nl + l)
        if composite:
            a = a.T
            if numeric and len(set(kinds)) > 1:
                return tuple(typed(a[l], kinds[l]) for l in range(nl))
        return typed(a, kinds[0])
    if len(pools) == 1:
        return arr(*pools[0])
    return tuple(arr(*p) for p in pools)


def _domain_points(config, ndraw, n, rng):
    """random draws inside the variables' domains, X[draw, sample]"""
    X = np.empty((ndraw, n))
    k = 0
    for vi, v in enumerate(config.var):
        nl = config.pool_width(vi)
        leaves = list(getattr(v, "vars", [v]))
        for s in range(config.maxdof[vi]):
            for l in range(nl):
                lf = leaves[l] if l < len(leaves) else leaves[0]
                lo, hi = float(getattr(lf, "lower", 0.0)), float(getattr(lf, "upper", 1.0))
                if hasattr(lf, "maxK"):                          # FermiK: momentum components
                    lo, hi = -lf.maxK / math.sqrt(lf.dim), lf.maxK / math.sqrt(lf.dim)
                if not (math.isfinite(lo) and math.isfinite(hi) and hi > lo):
                    lo, hi = 0.0, 1.0
                if hasattr(lf, "ninc") or hasattr(lf, "maxK") or not float(lo).is_integer():
                    X[k] = rng.uniform(lo, hi, n)
                else:                                            # Discrete: integer values
                    X[k] = rng.integers(int(lo), int(hi) + 1, n)
                k += 1
    return X


MAX_WAYS = 256   # ways through a closure's Python branches that are written out (a loop whose trip count depends on a draw has no bound)


_COSTLY = ("/", "pow", "atan2", "fmod") + _FUNCS      # operations worth a select per operand to be evaluated once instead of once per way


def _join(cond, a, b):
    """where(cond, a, b) for the values of two ways through a closure's branches, with the select pushed DOWN through what the two ways
    share: `c ? f(u, k) : f(v, k)` is written `f(c ? u : v, k)` -- the same number, bit for bit (f is pure and is applied to the selected
    operand), but the common operation is evaluated once.  `omega > 0 ? exp(-omega tau) / (1 + exp(-omega beta)) : exp(omega (beta -
    tau)) / (1 + exp(omega beta))` (test/bubble.jl:40-51) becomes two exponentials of selected arguments instead of four, which is what a
    hand-written body computes per lane; a product of a way-dependent factor with common ones keeps its common multiplications."""
    if isinstance(a, (CSym, complex, np.complexfloating)) or isinstance(b, (CSym, complex, np.complexfloating)):
        a, b = CSym.of(a), CSym.of(b)
        return CSym(_join(cond, a.re, b.re), _join(cond, a.im, b.im))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.frompyfunc(lambda p, q: _join(cond, p, q), 2, 1)(_boxed(a), _boxed(b))
    if a is b:
        return a
    if (isinstance(a, Sym) and isinstance(b, Sym) and a.op == b.op and len(a.args) == len(b.args)
            and a.op not in ("x", "rw", "ud", "const") and a.op not in _BOOL):
        diff = [i for i, (p, q) in enumerate(zip(a.args, b.args)) if p is not q and not (not isinstance(p, Sym) and not isinstance(q, Sym) and p == q)]
        sinkable = all(isinstance(a.args[i], Sym) and isinstance(b.args[i], Sym) and a.args[i].op not in _BOOL and b.args[i].op not in _BOOL
                       for i in diff)
        if diff and sinkable and (len(diff) == 1 or a.op in _COSTLY):
            args = list(a.args)
            for i in diff:
                args[i] = _join(cond, a.args[i], b.args[i])
            return a.t.node(a.op, *args)
    return where(cond, a, b)


def explore(t, run):
    """run() -> list of values, calling the closure on trace t's symbols; the closure may branch on sampled values.  Every way through
    its branches is run once (the k-th truth test of a run comes out as the script says, True beyond it) and the ways are joined into
    ONE list of values with selects: where(first test, values of the ways on which it held, values of the others) -- the C body then
    holds `c ? a : b` where the closure had `a if c else b`, like a hand-written body (and like Julia's inlined ternary).  Values may be
    Sym, CSym or numbers.  TraceError beyond MAX_WAYS ways."""
    ways = [0]

    def way(prefix):
        ways[0] += 1
        if ways[0] > MAX_WAYS:
            raise TraceError("more than %d ways through the closure's branches on sampled values (a loop that ends on a draw?)" % MAX_WAYS)
        t.script, t.conds, t.decided = list(prefix), [], []
        vals = list(run())
        conds, decided = list(t.conds), list(t.decided)
        for i in range(len(decided) - 1, len(prefix) - 1, -1):   # the tests this run met beyond its prefix, last first: each has an untaken way
            other = way(decided[:i] + [False])
            if len(other) != len(vals):
                raise TraceError("the closure returns %d values on one way through its branches and %d on another" % (len(vals), len(other)))
            same = lambda a, b: a is b or (not isinstance(a, (Sym, CSym, np.ndarray)) and not isinstance(b, (Sym, CSym, np.ndarray)) and a == b)
            vals = [a if same(a, b) else _join(conds[i], a, b) for a, b in zip(vals, other)]
        return vals
    try:
        return way([])
    finally:
        t.script, t.conds, t.decided = [], [], []


class _Weights(list):
    """the `weights` output vector of the in-place form `integrand(var, weights, config)` (vegas/montecarlo.jl:140-141) during a trace:
    N entries, zero until the closure stores into them (the reference hands the closure its reused buffer; an entry the closure
    never writes is a zero weight here)"""

    def __init__(self, n):
        super().__init__([0.0] * n)

    def __setitem__(self, i, v):
        if isinstance(i, slice):
            idx = range(*i.indices(len(self)))
            vals = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * len(idx)
            if len(vals) != len(idx):
                raise TraceError("weights[%r] = a sequence of another length" % (i,))
            for k, q in zip(idx, vals):
                list.__setitem__(self, k, q)
            return
        try:
            list.__setitem__(self, i, v)
        except (IndexError, TypeError) as e:
            raise TraceError("weights[%r]: %s (the vector has one entry per integrand, 0-based)" % (i, e))

    def fill(self, v):
        self[:] = v


def _call_form(fn, arg, config, N, indexed, inplace, weights):
    """one call of the closure in its form -> the list of N values (reference forms: main.jl:26-28)"""
    if indexed:
        return [fn(i, arg, config) for i in range(N)]
    if inplace:
        w = weights()
        fn(arg, w, config)
        return list(w)
    outs = fn(arg, config)
    if N == 1 and not isinstance(outs, (tuple, list)):
        outs = (outs,)
    return list(outs)


def trace_integrand(fn, config, indexed=False, check_points=32, name=None, parameters=True, inplace=False):
    """Run the closure once on symbolic draws and return the Integrand (HIP C++ body + userdata) that computes the same thing;
    TraceError if it cannot be written out or if the written-out body and the closure disagree at random points of the domain.
    `parameters`: captured floats become userdata slots (module docstring) -- the body does not depend on their values.

    fn(x, config) -> value | tuple of N values
    indexed=True: fn(idx, x, config) -> value, the reference's :mcmc form (mcmc/montecarlo.jl:34-36), idx 0-based
    inplace=True: fn(x, weights, config), the reference's `inplace = true` form (main.jl:26, vegas/montecarlo.jl:140-141,
        vegas_mc/updates.jl:67-70): the closure stores `weights[i] = value`, 0-based; what it returns is ignored
    Complex weights (Configuration(type=complex)): values may be complex combinations of draws (x[0] ** 2 * 1j); every weight is
    written out as its (re, im) pair, w[2 i] and w[2 i + 1]."""
    if indexed and inplace:
        raise ValueError("the :mcmc form integrand(idx, var, config) has no in-place variant (main.jl:26-28)")
    if parameters:
        try:
            return _trace_integrand(fn, config, indexed, check_points, name, True, inplace)
        except TraceError:
            pass   # (a branch on a captured float, a parameter where Python wants a number): once more with the captured values as literals
    return _trace_integrand(fn, config, indexed, check_points, name, False, inplace)


def _trace_integrand(fn, config, indexed, check_points, name, parameters, inplace=False):
    pools, ndraw = _pools(config)
    t = _Trace()
    arg = _argument(pools, lambda k: t.node("x", k), pad=lambda: t.const(0.0))
    N = config.N
    nc = getattr(config, "ncomp", 1)
    sfn = _parametrized(fn, t, floats=parameters)          # (captured arrays can be indexed with a sampled value either way: _Table)
    tconfig = _trace_config(config, t, floats=parameters)
    def run():
        outs = _call_form(sfn, arg, tconfig, N, indexed, inplace, lambda: _Weights(N))
        if len(outs) != N:
            raise TraceError("the integrand must return one value per integrand (%d), got %d" % (N, len(outs)))
        return [o.reshape(-1)[0] if isinstance(o, np.ndarray) and o.size == 1 else o for o in outs]
    try:
        outs = explore(t, run)
    except TraceError:
        raise
    except Exception as e:   # whatever else the closure does with a symbol that a float would have survived
        raise TraceError("%s: %s" % (type(e).__name__, e))
    syms = []
    for o in outs:
        if nc == 2:
            o = CSym.of(o)
            syms += [t.lift(o.re), t.lift(o.im)]
            continue
        if isinstance(o, CSym) or isinstance(o, (complex, np.complexfloating)):
            raise TraceError("a complex weight in a real configuration (Configuration(type=complex) makes the weights complex)")
        if not isinstance(o, Sym):
            o = t.const(o)
        syms.append(o)
    slots, ud = hoist(syms, t.params)
    ud, tables = list(ud), {}
    for n in _reachable(syms, stop=slots):                  # the tables the body looks into follow the parameters in ud[]
        if n.op == "table" and n.args[0] not in tables:
            tables[n.args[0]] = len(ud)
            ud += [float(v) for v in t.tables[n.args[0]].reshape(-1)]
    body = emit(syms, leaves=slots, tables=tables)
    if check_points:
        rng = np.random.default_rng(12345)
        X = _domain_points(config, ndraw, check_points, rng)
        ref = np.empty((N * nc, check_points))
        try:
            with np.errstate(all="ignore"):
                for p in range(check_points):   # the closure on one sample at a time: plain floats where the trace had symbols
                    num = _argument(pools, lambda k: X[k, p], numeric=True)
                    r = _call_form(fn, num, config, N, indexed, inplace, lambda: np.zeros(N, dtype=complex if nc == 2 else float))
                    for i in range(N):
                        if nc == 2:
                            z = complex(np.asarray(r[i], dtype=np.complex128).reshape(-1)[0])
                            ref[2 * i, p], ref[2 * i + 1, p] = z.real, z.imag
                        else:
                            ref[i, p] = float(np.asarray(r[i], dtype=np.float64).reshape(-1)[0])
        except Exception as e:
            raise TraceError("the closure does not run on numeric draws (%s: %s)" % (type(e).__name__, e))
        got = evaluate(syms, X, params=t.params, tables=t.tables)
        for i in range(N * nc):
            r = ref[i]
            ok = np.isfinite(r) & np.isfinite(got[i])
            if not np.array_equal(np.isfinite(r), np.isfinite(got[i])) or not np.allclose(got[i][ok], r[ok], rtol=1e-10, atol=1e-290):
                raise TraceError("the traced expression and the closure disagree on integrand %d: the closure is not a pure "
                                 "function of its draws (hidden state, a branch the trace did not see, numpy arithmetic on "
                                 "comparisons that means something else than the same arithmetic on 0.0 / 1.0)" % (i // nc))
    return Integrand(body, ud or None, name=name or getattr(fn, "__name__", "traced"))


class _Obs(np.ndarray):
    """one observable during the trace of a measure: an object array that starts at zero and remembers what is added to its entries;
    `obs[i][k] += value` with a SAMPLED k (a Discrete draw) is recorded on the trace as an add to a bin chosen at run time"""

    def __new__(cls, n, t, oi, zero):
        obj = np.empty(n, dtype=object).view(cls)
        obj[:] = zero
        obj._t, obj._oi, obj._zero = t, oi, zero
        return obj

    def __array_finalize__(self, obj):
        self._t = getattr(obj, "_t", None)
        self._oi = getattr(obj, "_oi", None)
        self._zero = getattr(obj, "_zero", None)

    @staticmethod
    def _sampled(i):
        return isinstance(i, Sym) or (isinstance(i, tuple) and any(isinstance(q, Sym) for q in i))

    def __getitem__(self, i):
        if self._sampled(i):
            return self._zero                       # (what `+=` reads before it adds)
        return np.ndarray.__getitem__(self, i)

    def __setitem__(self, i, v):
        if self._sampled(i):
            if self._t is None or self._oi is None or self.base is not None and self.base.shape != self.shape:
                raise TraceError("a sampled index into a view of an observable")
            self._t.dynamic.append((self._oi, self._flat_bin(i), v))
            return
        np.ndarray.__setitem__(self, i, v)

    def _flat_bin(self, i):
        """the row-major flat bin of `obs[i][a, b]` (sampled values and numbers on ALL axes), -1 where an entry is off its axis"""
        if isinstance(i, Sym):
            i = (i,)
        if len(i) != self.ndim or not all(isinstance(q, (Sym, int, np.integer)) for q in i):
            raise TraceError("an observable indexed with a sampled value needs one index per axis (numbers or sampled values)")
        if self.ndim == 1:
            return i[0]                             # (the range check is the written-out body's)
        flat, inside = 0.0, []
        for n, q in zip(self.shape, i):
            if isinstance(q, Sym):
                inside.append((q, n))
                flat = flat * float(n) + q
            else:
                if not -n <= int(q) < n:
                    raise TraceError("index %d is out of bounds for an axis of %d" % (int(q), n))
                flat = flat * float(n) + float(int(q) % n)
        for q, n in inside:
            flat = where(q < 0.0, -1.0, where(q > float(n - 1), -1.0, flat))
        return flat


def trace_measure(fn, config, indexed=False, check_points=32):
    """The `measure` closure (vegas/montecarlo.jl:156-161; five-argument :mcmc form mcmc/montecarlo.jl:166-169) written out as a
    device Measure: the closure is run once with symbolic draws, symbolic relative weights and `obs` arrays that remember what is
    added to them; `obs[i][k] += expr` becomes `obs_add(flat k, expr)`.  Both the per-sample form of the reference
    (`obs[0][0] += weights[0]`) and the batch-vectorised form of HostMeasure (`obs[0][0] += weights[0].sum()`,
    `weights[0][x[0] < 0.5].sum()`) trace.  TraceError if it cannot be written out or disagrees with the closure at random points.

    fn(x, obs, weights, config)     (indexed=True: fn(idx, x, obs, weight, config), idx 0-based)"""
    nc = getattr(config, "ncomp", 1)   # 2: ComplexF64 weights and observables, every one an (re, im) pair of slots (rw[2 i], rw[2 i + 1]; obs likewise)
    pools, ndraw = _pools(config)
    N = config.N
    t = _Trace()
    arg = _argument(pools, lambda k: t.node("x", k), pad=lambda: t.const(0.0))
    zero = t.const(0.0)

    def fresh():
        return [_Obs(shape, t, oi, zero) for oi, shape in enumerate(config.obs_shape)]

    shapes = []

    def flat(obs):
        out = []
        for o in obs:
            for v in np.asarray(o, dtype=object).reshape(-1):
                if nc == 2:
                    z = CSym.of(v)
                    out += [t.lift(z.re), t.lift(z.im)]
                elif isinstance(v, (CSym, complex, np.complexfloating)):
                    raise TraceError("a complex observable in a real configuration")
                else:
                    out.append(v if isinstance(v, Sym) else t.const(v))
        # `obs[i][sampled index] += value` (a histogram over a Discrete draw: docs/src/index.md "Measure Histogram", test/bubble.jl:87):
        # index and value follow the fixed slots, one group per such statement; every way through the measure must make the same ones
        for oi, idx, v in t.dynamic:
            out.append(idx)
            if nc == 2:
                z = CSym.of(v)
                out += [t.lift(z.re), t.lift(z.im)]
            elif isinstance(v, (CSym, complex, np.complexfloating)):
                raise TraceError("a complex observable in a real configuration")
            else:
                out.append(t.lift(v))
        shapes[-1].append(tuple(oi for oi, _, _ in t.dynamic))
        return out

    def weight(i):
        return CSym(t.node("rw", 2 * i), t.node("rw", 2 * i + 1)) if nc == 2 else t.node("rw", i)
    def run_one(i):   # (every way through the measure's Python branches is run on fresh observables, explore())
        shapes.append([])
        def run():
            obs = fresh()
            t.dynamic = []
            if i is None:
                fn(arg, obs, [weight(k) for k in range(N)], config)
            else:
                fn(i, arg, obs, weight(i), config)
            return flat(obs)
        return explore(t, run)
    try:
        per = [run_one(i) for i in range(N)] if indexed else [run_one(None)]
    except TraceError:
        raise
    except Exception as e:
        raise TraceError("%s: %s" % (type(e).__name__, e))
    nobs = sum(config.obs_len) * nc
    if any(len(set(sh)) > 1 for sh in shapes):
        raise TraceError("the measure adds to sampled bins on some ways through its branches and not on others")
    dyns = [sh[0] if sh else () for sh in shapes]  # per call form: the observable of every `obs[i][sampled index] += value`
    if any(len(p) != nobs + len(dyn) * (1 + nc) for p, dyn in zip(per, dyns)):
        raise TraceError("the measure changed the shape of obs")
    obs_off = [sum(config.obs_len[:oi]) * nc for oi in range(len(config.obs_len))]
    parts = []
    for i, adds in enumerate(per):
        ks = [k for k, v in enumerate(adds[:nobs]) if not _is_const(v, 0.0)]
        dyn = dyns[i]

        def sink(j, ref, ks=ks, i=i, dyn=dyn):
            if j < len(ks):
                return "obs_add(%d, %s);" % (ks[j], ref)
            d, q = divmod(j - len(ks), 1 + nc)
            if q == 0:
                return "const int mci_k%d_%d = (int)(%s);" % (i, d, ref)
            oi = dyn[d]
            at = "mci_k%d_%d" % (i, d) if nc == 1 else "2 * mci_k%d_%d + %d" % (i, d, q - 1)
            return "if (mci_k%d_%d >= 0 && mci_k%d_%d < %d) obs_add(%d + %s, %s);" % (i, d, i, d, config.obs_len[oi], obs_off[oi], at, ref)
        body = emit([adds[k] for k in ks] + list(adds[nobs:]), sink=sink) if ks or dyn else ""
        if indexed:
            # (under :vegas / :vegasmc every integrand's weight is measured: idx = -1; an :mcmc chain measures the one it sits on)
            body = "if (idx < 0 || idx == %d) {\n%s\n}" % (i, body) if body else ""
        parts.append(body)
    if check_points:
        rng = np.random.default_rng(54321)
        X = _domain_points(config, ndraw, check_points, rng)
        R = rng.standard_normal((N * nc, check_points))
        cdt = complex if nc == 2 else float

        def wnum(i, p):
            return np.complex128(complex(R[2 * i, p], R[2 * i + 1, p])) if nc == 2 else np.float64(R[i, p])   # (numpy scalars: `.sum()` and masks work on them)

        def oflat(obs):
            v = np.concatenate([np.asarray(o, dtype=cdt).reshape(-1) for o in obs])
            return np.stack([v.real, v.imag], axis=1).reshape(-1) if nc == 2 else v.astype(np.float64)
        for p in range(check_points):
            num = _argument(pools, lambda k: X[k, p], numeric=True)   # one record as numpy scalars: `.sum()`, masks and plain `+=` all work on them
            try:
                with np.errstate(all="ignore"):
                    if indexed:
                        refs = []
                        for i in range(N):
                            obs = [np.zeros(shape, dtype=cdt) for shape in config.obs_shape]
                            fn(i, num, obs, wnum(i, p), config)
                            refs.append(oflat(obs))
                    else:
                        obs = [np.zeros(shape, dtype=cdt) for shape in config.obs_shape]
                        fn(num, obs, [wnum(i, p) for i in range(N)], config)
                        refs = [oflat(obs)]
            except Exception as e:
                raise TraceError("the measure does not run on numeric records (%s: %s)" % (type(e).__name__, e))
            for adds, ref, dyn in zip(per, refs, dyns):
                ev = [float(v[0]) for v in evaluate(adds, X[:, p:p + 1], R[:, p:p + 1])]
                got = np.array(ev[:nobs])
                for d, oi in enumerate(dyn):
                    k = int(ev[nobs + d * (1 + nc)])
                    if 0 <= k < config.obs_len[oi]:
                        for q in range(nc):
                            got[obs_off[oi] + nc * k + q] += ev[nobs + d * (1 + nc) + 1 + q]
                if not np.allclose(got, ref, rtol=1e-10, atol=1e-290, equal_nan=True):
                    raise TraceError("the traced measure and the closure disagree: the closure is not a pure function of its records")
    # (an empty body would mean "the default measure" to the library: a closure that adds nothing is written out as a no-op)
    return Measure("\n".join(q for q in parts if q) or "(void)0;")
