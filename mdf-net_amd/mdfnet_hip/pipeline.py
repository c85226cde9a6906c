"""Items in flight: n HIP streams, independent work items issued round-robin.

One 5-view item leaves the card partly idle (kernel tails, the small per-pixel heads, the tiny control-plane copies); a
second item on another stream fills those gaps: +11 % (2 streams) / +15 % (3) views/s at cfg2, outputs bit-identical
(scripts/bench_streams.py).  Used by eval.py (whole-scan runs) and bench.py.  The conv kernels keep one scheduler slot per
stream (csrc/conv_lds.hip), every other kernel is stateless."""
import collections
import os

import torch

# items in flight used by BOTH the eval driver (eval.py:run_eval) and bench.py, so the benchmark times what eval issues
DEFAULT_IN_FLIGHT = int(os.environ.get("MDF_IN_FLIGHT", "3"))


class InFlight:
    def __init__(self, device, n=2, done=None):
        self.device = torch.device(device)
        self.n = max(1, int(n))
        self.done = done
        self.queue = collections.deque()
        self.count = 0
        self.cuda = self.device.type == "cuda"
        if not self.cuda:
            self.n = 1                                 # CPU (stock-op) runs: plain synchronous calls
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.n)] if self.n > 1 else [None]

    def submit(self, fn, tag=None, keep=None):
        """Enqueue `fn()` (kernel launches only) on the next stream; returns as soon as it is issued.  `keep`: objects that
        must stay alive until the item has finished (its input tensors)."""
        s = self.streams[self.count % self.n]
        self.count += 1
        if not self.cuda:
            out = fn()
            if self.done is not None:
                self.done(tag, out)
            return
        if s is None:
            out = fn()
            ev = torch.cuda.Event()
            ev.record()
        else:
            s.wait_stream(torch.cuda.current_stream(self.device))       # inputs were produced on the caller's stream
            with torch.cuda.stream(s):
                out = fn()
                ev = torch.cuda.Event()
                ev.record(s)
        self.queue.append((tag, out, ev, keep))
        pending = self.n if self.n > 1 else 0      # one stream: hand the result over at once (synchronous, as before)
        while len(self.queue) > pending:
            self._pop()

    def _pop(self):
        tag, out, ev, keep = self.queue.popleft()
        ev.synchronize()
        if self.done is not None:
            self.done(tag, out)
        return tag, out

    def drain(self):
        while self.queue:
            self._pop()
