"""A small reverse-mode tape with differentiable backward passes.

The reference relies on torch.autograd for ``loss.backward()`` (srgan.py:264,280-295,304) and for the
double backward of the gradient penalty (``autograd.grad(..., create_graph=True)``, srgan.py:368-370).
Here every differentiable operation is a HIP kernel launch recorded on this tape; each operation's
backward is itself written with tape operations, so differentiating a gradient (the penalty) needs nothing
special.  torch is used only as the device allocator behind ``Var.data``.
"""
import contextlib

import torch

_grad_enabled = True


def grad_enabled():
    return _grad_enabled


@contextlib.contextmanager
def no_grad():
    global _grad_enabled
    previous, _grad_enabled = _grad_enabled, False
    try:
        yield
    finally:
        _grad_enabled = previous


@contextlib.contextmanager
def enable_grad():
    global _grad_enabled
    previous, _grad_enabled = _grad_enabled, True
    try:
        yield
    finally:
        _grad_enabled = previous


class Node:
    """One recorded operation: its inputs and a function mapping the output gradient to input gradients."""
    __slots__ = ('inputs', 'input_requires', 'backward', 'name')

    def __init__(self, inputs, backward, name=''):
        self.inputs = inputs
        # Which inputs wanted gradients WHEN THE OPERATION RAN (e.g. the discriminator's parameters are frozen
        # while the generator loss is built; un-freezing them afterwards must not resurrect those gradients).
        self.input_requires = tuple(v is not None and v.requires_grad for v in inputs)
        self.backward = backward
        self.name = name


class Var:
    """A device tensor (contiguous fp32; or 16-bit in the blocked layout, see ``blocked16``) with an optional producer node.

    Leaves created from parameters carry ``grad_buffer``: a view of the network's flat gradient arena into
    which ``backward`` accumulates."""
    __slots__ = ('data', 'node', 'requires_grad', 'grad_buffer', 'grad', 'meta', '__weakref__')

    def __init__(self, data, requires_grad=False, node=None):
        self.data = data
        self.requires_grad = requires_grad
        self.node = node
        self.grad_buffer = None
        self.grad = None
        self.meta = None          # blocked16.Blocked for a 16-bit tensor in the blocked layout (None: plain fp32)

    @property
    def shape(self):
        return tuple(self.data.shape)

    def numel(self):
        return self.data.numel()

    def detach(self):
        return Var(self.data)

    def item(self):
        return self.data.item()

    def cpu(self):
        return self.data.detach().cpu()

    def size(self, dim=None):
        return self.data.size() if dim is None else self.data.size(dim)

    def __repr__(self):
        return f'Var(shape={self.shape}, requires_grad={self.requires_grad}, op={self.node.name if self.node else None})'


def _topological_order(root, relevant):
    """Nodes reachable from ``root`` through vars in ``relevant`` (or all requiring grad), consumers first."""
    order, visited, stack = [], set(), [(root, False)]
    while stack:
        var, expanded = stack.pop()
        if expanded:
            order.append(var)
            continue
        if id(var) in visited or var.node is None:
            continue
        visited.add(id(var))
        stack.append((var, True))
        for parent, required in zip(var.node.inputs, var.node.input_requires):
            if required and (relevant is None or id(parent) in relevant):
                stack.append((parent, False))
    order.reverse()
    return order


def _relevant_set(root, inputs):
    """ids of vars on a path from ``root`` down to one of ``inputs``."""
    wanted = {id(v) for v in inputs}
    memo = {}
    stack = [(root, False)]
    while stack:
        var, expanded = stack.pop()
        key = id(var)
        if expanded:
            memo[key] = key in wanted or any(p is not None and memo.get(id(p), False) for p in var.node.inputs)
            continue
        if key in memo:
            continue
        if var.node is None:
            memo[key] = key in wanted
            continue
        memo[key] = key in wanted     # provisional (cycle-free graph: only used before expansion completes)
        stack.append((var, True))
        for parent, required in zip(var.node.inputs, var.node.input_requires):
            if required and id(parent) not in memo:
                stack.append((parent, False))
    return {key for key, value in memo.items() if value}


def _frontiers(order, flat):
    """frontier[i] = the largest end offset (in elements of the flat gradient buffer ``flat``) that node ``order[i]`` or
    any LATER node of the sweep can still write: once the sweep has processed nodes 0..i-1, every element of ``flat``
    at or after frontier[i] is final."""
    base, count = flat.data_ptr(), flat.numel()
    frontier = [0] * (len(order) + 1)
    for i in range(len(order) - 1, -1, -1):
        reach = frontier[i + 1]
        node = order[i].node
        for parent, required in zip(node.inputs, node.input_requires):
            if not required or parent.node is not None or parent.grad_buffer is None:
                continue
            offset = (parent.grad_buffer.data_ptr() - base) // 4
            if 0 <= offset < count:
                reach = max(reach, offset + parent.grad_buffer.numel())
        frontier[i] = reach
    return frontier


_sweep_end_hooks = []     # callables run (once) when the outermost running sweep ends, or earlier by ``run_sweep_end_hooks``
_sweep_depth = 0


def at_sweep_end(hook):
    """Run ``hook()`` when the backward sweep in progress has enqueued its last node -- e.g. the join of a side stream that
    carries weight-gradient launches nothing inside the sweep depends on.  Outside a sweep the hook runs at once."""
    if _sweep_depth == 0:
        hook()
    else:
        _sweep_end_hooks.append(hook)


def run_sweep_end_hooks():
    while _sweep_end_hooks:
        _sweep_end_hooks.pop(0)()


_accumulating = False     # inside a plain backward sweep (no ``inputs``, not recorded)
_exclusive = False        # the gradient handed to the node being swept is referenced by nobody else


def incoming_gradient_is_exclusive():
    """True when the ``g`` of the ``node.backward(g, needs)`` call in progress was produced for this node alone (one
    producer slot or a fresh sum, in a sweep that does not record): the node may then overwrite it in place."""
    return _exclusive


def accumulates_into(var):
    """True while a plain ``backward`` sweep runs and ``var`` is a parameter leaf: the operation's backward may then add
    its gradient straight into ``var.grad_buffer`` (the arena) with the kernel's own accumulate mode and return None for
    it, instead of materialising a temporary that the sweep adds afterwards (one launch and one tensor per parameter)."""
    return _accumulating and var is not None and var.node is None and var.grad_buffer is not None and var.requires_grad


def backward(root, grad=None, inputs=None, create_graph=False, retain_graph=None, grad_ready=None):
    """Reverse sweep from ``root``.

    ``grad_ready`` (a ``parallel.GradientExchange`` over a network's flat gradient arena; only for the LAST backward
    pass of a step into that arena): after every node the sweep reports from which arena offset on no later node can
    write, and the exchange starts all-reducing that tail while the sweep continues.

    Without ``inputs`` gradients are accumulated into the ``grad_buffer`` (parameters) or ``grad`` (other
    leaves) of every leaf that requires grad, like ``Tensor.backward``.  With ``inputs`` the gradients of
    exactly those vars are returned and nothing is accumulated, like ``torch.autograd.grad``; only the part
    of the graph between ``root`` and ``inputs`` is differentiated (so e.g. no weight gradients are computed
    for the gradient-penalty's inner gradient).  ``create_graph`` records the backward pass itself.
    A requested input ENDS the differentiated sub-graph: the sweep does not continue through its producer.  For inputs
    that do not depend on each other (every caller: the interpolates; x, gamma, beta of one op) that is
    ``torch.autograd.grad``; with nested inputs -- one a function of another -- the outer one's gradient would lack the
    path through the inner one: do not pass such a list.
    """
    from . import functional as F   # local import: functional builds on this module
    if retain_graph is None:
        retain_graph = create_graph
    if not root.requires_grad:
        raise RuntimeError('backward() on a var that does not require grad')
    if grad is None:
        if root.numel() != 1:
            raise RuntimeError('grad can be implicitly created only for scalar outputs')
        grad = F.full_like(root, 1.0)
    relevant = _relevant_set(root, inputs) if inputs is not None else None
    order = _topological_order(root, relevant)
    grads = {id(root): grad}
    shared = set()            # ids of gradient vars that some other consumer may still read
    context = enable_grad() if create_graph else no_grad()
    results = {}
    wanted = {id(v): v for v in inputs} if inputs is not None else {}
    frontier = _frontiers(order, grad_ready.flat) if grad_ready is not None else None
    global _accumulating, _exclusive, _sweep_depth
    previous, _accumulating = _accumulating, (inputs is None and not create_graph)
    _sweep_depth += 1
    try:
        with context:
            if id(root) in wanted:
                results[id(root)] = grad
            for position, var in enumerate(order):
                if frontier is not None:
                    run_sweep_end_hooks()     # side-stream launches write the arena too: joined before a tail is declared final
                    grad_ready.ready_from(frontier[position])
                g = grads.pop(id(var), None)
                if g is None or id(var) in wanted:
                    # (a requested input is the END of the differentiated sub-graph even when it has a producer: running that
                    # producer's backward -- with no input wanting a gradient -- would be a whole wasted pass, e.g. a dense
                    # block's data-gradient chain behind a fused op that re-evaluates itself for a recorded backward)
                    continue
                node = var.node
                needs = tuple(required and (relevant is None or id(p) in relevant)
                              for p, required in zip(node.inputs, node.input_requires))
                _exclusive = not create_graph and not retain_graph and id(g) not in shared and g is not grad
                try:
                    input_grads = node.backward(g, needs)
                finally:
                    _exclusive = False
                # Which of the returned gradients may a consumer overwrite in place?  Not one that is handed to several
                # inputs, passed through, or that SHARES MEMORY with another returned gradient or with the incoming one
                # (distinct Var wrappers over one storage, row slices at an offset: judged by storage + byte range).
                returned = [pg for pg in input_grads if pg is not None]
                handed_out = [id(pg) for pg in returned]
                spans = [_span(pg.data) for pg in returned]
                incoming = _span(g.data)
                for index, pg in enumerate(returned):
                    aliased = handed_out.count(id(pg)) > 1 or pg is g or \
                        (id(g) in shared and _overlap(spans[index], incoming)) or \
                        any(other != index and returned[other] is not pg and _overlap(spans[index], spans[other])
                            for other in range(len(returned)))
                    if aliased:
                        shared.add(id(pg))
                for parent, need, pg in zip(node.inputs, needs, input_grads):
                    if not need or pg is None:
                        continue
                    if pg.shape != parent.shape:
                        raise RuntimeError(f'{node.name}: gradient shape {pg.shape} != input shape {parent.shape}')
                    key = id(parent)
                    if parent.node is None:                      # leaf
                        if inputs is not None:
                            if key in wanted:
                                results[key] = pg if key not in results else F.add(results[key], pg)
                        elif parent.grad_buffer is not None:
                            F.accumulate_(parent.grad_buffer, pg)
                        else:
                            parent.grad = pg if parent.grad is None else F.add(parent.grad, pg)
                    else:
                        if key in wanted:
                            results[key] = pg if key not in results else F.add(results[key], pg)
                            shared.add(id(pg))                 # the caller gets it back
                        grads[key] = pg if key not in grads else F.add(grads[key], pg)
                if not retain_graph:
                    node.backward = _released
                    node.inputs = ()
    finally:
        _accumulating = previous
        _sweep_depth -= 1
        if _sweep_depth == 0:
            run_sweep_end_hooks()
    if frontier is not None:
        grad_ready.ready_from(0)
    if inputs is not None:
        return [results.get(id(v)) for v in inputs]
    return None


def _span(tensor):
    """(storage address, first byte, one past the last byte the tensor can touch): the extent follows the sizes and strides,
    so a strided view (a channel slice ``buffer[:, a:b]``) covers everything from its first to its last element."""
    first = tensor.data_ptr()
    if tensor.numel() == 0:
        return tensor.untyped_storage().data_ptr(), first, first
    extent = 1 + sum((size - 1) * abs(stride) for size, stride in zip(tensor.shape, tensor.stride()))
    return tensor.untyped_storage().data_ptr(), first, first + extent * tensor.element_size()


def _overlap(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


def _released(*_):
    raise RuntimeError('this part of the graph was already back-propagated through and freed')
