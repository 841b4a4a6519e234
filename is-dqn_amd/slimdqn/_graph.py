"""hipGraph capture of the replay-sample -> Bellman-update step.

One step is ~30 kernel launches on two HIP streams; launched eagerly the host spends ~380 us per step on
launch API calls, about as long as the GPU needs to run them.  ``GraphedUpdate`` captures `S` consecutive
steps (index rows -> [sum-tree query] -> row gather -> learn_on_batch -> [priority write-back]) once, with
static buffers, and replays them: one graph launch per S steps.

The draws stay on the host (numpy PCG64, the reference's stream): before a replay the next S rows of
pre-drawn dense indices (uniform) or unit draws (prioritized) are copied into the static input block.
"""
from __future__ import annotations

import torch

from slimdqn import _hip


class GraphedUpdate:
    def __init__(self, rb, eng, prioritized: bool, steps_per_graph: int = 8, writeback: bool | None = None, learn=None):
        """``prioritized``: the sampler is a sum tree (the query is part of the graph); ``writeback``: sqrt(mean_k td) of
        every step goes back into the tree (default: whenever prioritized); ``learn``: the learn call of one step on a C batch
        (default ``eng.learn_on_batch``; DQN passes its target-parameter form, dqn.py:59-72)."""
        self.rb, self.eng, self.prioritized, self.S = rb, eng, prioritized, steps_per_graph
        self._learn = eng.learn_on_batch if learn is None else learn
        self.writeback = prioritized if writeback is None else (writeback and prioritized)
        dev, B, s2 = eng.device, eng.batch_size, 2 * rb._stack_size
        self.B = B
        rb._flush()
        self.block = torch.zeros(self.S, B, dtype=torch.float64 if prioritized else torch.int32, device=dev)
        self.indices = torch.zeros(B, dtype=torch.int32, device=dev)
        # Uniform sampling: the S steps of a replay sample an unchanged buffer with indices drawn beforehand, so their S
        # row gathers are ONE launch over S*B rows at the head of the graph (one 6 us kernel at the head of a step
        # less); every step then has its own static batch.  Prioritized sampling depends on the previous step's
        # priority write-back and keeps one gather per step into a single batch.
        # Prioritized: two batch slots alternate, because the write-back of step s and the draw + gather of step s+1 run on a
        # second stream under the backward pass of step s (learn_on_batch records `priorities_ready` for it).
        n_b = 2 if prioritized else self.S
        if prioritized:
            self.indices = torch.zeros(2, B, dtype=torch.int32, device=dev)
            self._sampling_stream = torch.cuda.Stream(dev)
            self._prio_ready = [torch.cuda.Event() for _ in range(2)]
            for e in self._prio_ready:
                e.record(self._sampling_stream)  # creates the handle the C ABI takes
        self.frame_ids = torch.zeros(n_b, B, s2, dtype=torch.int32, device=dev)
        self.action = torch.zeros(n_b, B, dtype=torch.int32, device=dev)
        self.reward = torch.zeros(n_b, B, dtype=torch.float32, device=dev)
        self.terminal = torch.zeros(n_b, B, dtype=torch.uint8, device=dev)
        # vector observations (the fc torso, LunarLander): the replay stores the bytes of every float32 observation as one 1 x 4d "frame";
        # the captured step materialises the B state / next-state rows of a batch into static buffers the engine reads as (B, d) float32
        self.fc = eng.architecture_type == "fc"
        if self.fc:
            if not rb._obs_float or rb._stack_size != 1:
                raise NotImplementedError("captured fc step: float32 vector observations with stack_size 1")
            self.obs = torch.zeros(2, n_b * B, rb._w, dtype=torch.uint8, device=dev)  # [state | next_state][slot * B + b][4 d]
        self._frames_ptr = rb._frames.data_ptr()
        self._make_batches()
        # Every captured step takes the weight mirror as it is; run() rebuilds it in front of the replay (one eager 8 us launch) only
        # when the engine's bookkeeping says something wrote the parameters since the last call that left it current (consecutive
        # replays and acting in between do not: _engine.py _mirror_is_current).  (A second capture that rebuilds at its head was the
        # first form of this; it crashed hipGraphLaunch in hip::Graph::UpdateStreams in some test orders.  Cause, DESIGN.md
        # "hipGraphLaunch fault": at launch the runtime fills the graph's stream array from the executable's own pool of
        # parallel streams, skips every pool stream that maps to the launch stream's queue, and indexes the pool WITHOUT a bound:
        # one skipped entry too many and it reads a Stream* past the end of the vector.  Which pool stream aliases the launch
        # stream depends on how many streams the process created before the instantiation -- the test order.  The agents therefore
        # keep ONE live executable each, destroy it before capturing a successor, and never re-instantiate in steady state.)
        self.graph = None
        self._capture()

    def destroy(self) -> None:
        """Release the executable graph (and the runtime's parallel streams it owns) after everything it enqueued has finished.
        An agent calls this on the update object it replaces BEFORE it captures the successor."""
        if self.graph is not None:
            torch.cuda.synchronize(self.eng.device)
            self.graph.reset()
            self.graph = None

    def _make_batches(self) -> None:
        rb, eng = self.rb, self.eng
        # every step of a replay finds the weight mirror current: steps 2..S follow a learn step of the same graph, step 1 follows
        # run()'s check
        ev = lambda i: self._prio_ready[i] if self.prioritized else None
        if self.fc:
            B = self.B
            rows = lambda half, i: self.obs[half, i * B:(i + 1) * B].view(torch.float32)
            self.chained = [
                eng.make_batch(state=rows(0, i), next_state=rows(1, i), action=self.action[i], reward=self.reward[i], terminal=self.terminal[i],
                               mirror_current=True, priorities_ready=ev(i))
                for i in range(self.frame_ids.shape[0])
            ]
            return
        self.chained = [
            eng.make_batch(frames=rb._frames, frame_stride=rb._hw, frame_ids=self.frame_ids[i], action=self.action[i],
                           reward=self.reward[i], terminal=self.terminal[i], mirror_current=True, priorities_ready=ev(i))
            for i in range(self.frame_ids.shape[0])
        ]

    def _gather(self, idx, n_rows: int, slot: int) -> None:
        rb = self.rb
        _hip.check(
            rb._lib.isdqn_replay_gather_rows(
                _hip.ptr(rb._d_elem_frames), _hip.ptr(rb._d_elem_action), _hip.ptr(rb._d_elem_reward),
                _hip.ptr(rb._d_elem_terminal), rb._stack_size, _hip.ptr(rb._d_index_to_slot), _hip.ptr(idx), n_rows,
                _hip.ptr(self.frame_ids[slot]), _hip.ptr(self.action[slot]), _hip.ptr(self.reward[slot]),
                _hip.ptr(self.terminal[slot]), _hip.stream_ptr(self.eng.device),
            ),
            "isdqn_replay_gather_rows",
        )

    def _materialize(self, slot: int, n_rows: int) -> None:
        """fc: the observation rows of ``n_rows`` gathered transitions, from batch slot ``slot`` on."""
        rb, B = self.rb, self.B
        _hip.check(
            rb._lib.isdqn_replay_materialize(
                _hip.ptr(rb._frames), rb._hw, rb._h, rb._w, rb._stack_size, _hip.ptr(self.frame_ids[slot]), n_rows,
                _hip.ptr(self.obs[0, slot * B:]), _hip.ptr(self.obs[1, slot * B:]), _hip.stream_ptr(self.eng.device),
            ),
            "isdqn_replay_materialize",
        )

    def _steps(self) -> None:
        """The S steps of one replay, enqueued on the current stream.  The first node rebuilds the weight mirror from the parameters
        as they are when the replay starts (whoever wrote them, however: _engine.py "weight-mirror bookkeeping"); the steps behind
        it trust it -- each follows a learn step of the same replay, whose optimizer wrote both forms."""
        rb, eng = self.rb, self.eng
        if not eng.trust_mirror:
            eng.rebuild_mirror()
        if self.prioritized:
            tree = rb._sampling_distribution._sum_tree
            main, sampling = torch.cuda.current_stream(eng.device), self._sampling_stream
            tree.query_device(self.block[0], out=self.indices[0], unit=True)
            self._gather(self.indices[0], self.B, 0)
            for s in range(self.S):
                slot = s & 1
                if self.fc:
                    self._materialize(slot, self.B)
                self._learn(self.chained[slot])
                # under the rest of this step (backward, Adam): priorities of step s into the tree, then the draw and the row
                # gather of step s+1 -- the order the reference's loop has (update, then sample)
                sampling.wait_event(self._prio_ready[slot])
                with torch.cuda.stream(sampling):
                    if self.writeback:
                        rb._sampling_distribution.update_device(self.indices[slot], eng.priorities)
                    if s + 1 < self.S:
                        tree.query_device(self.block[s + 1], out=self.indices[1 - slot], unit=True)
                        self._gather(self.indices[1 - slot], self.B, 1 - slot)
                main.wait_stream(sampling)
        else:
            self._gather(self.block, self.S * self.B, 0)  # rows of all S steps: the [S][B] buffers are contiguous
            if self.fc:
                self._materialize(0, self.S * self.B)
            for s in range(self.S):
                self._learn(self.chained[s])

    def _capture(self) -> None:
        # warm-up on a side stream (lazy one-time setup inside the library must not happen during capture)
        side = torch.cuda.Stream(self.eng.device)
        side.wait_stream(torch.cuda.current_stream(self.eng.device))
        state = [t.clone() for t in (self.eng.params, self.eng.adam_m, self.eng.adam_v, self.eng.adam_count, self.eng.losses_accum)]
        # (prioritized: the warm-up also writes priorities back -- with the zero-filled draw block -- so the whole tree
        # state is saved: nodes, max_recorded_priority, which every later add() reads, and the latched status word)
        tree = self.rb._sampling_distribution._sum_tree if self.prioritized else None
        tree_state = [t.clone() for t in (tree._nodes_dev, tree._max_dev, tree._status)] if tree is not None else None
        with torch.cuda.stream(side):
            self.eng.rebuild_mirror()
            self._steps()
        torch.cuda.current_stream(self.eng.device).wait_stream(side)
        torch.cuda.synchronize(self.eng.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._steps()
        # the warm-up steps must not count: restore the training state
        for dst, src in zip((self.eng.params, self.eng.adam_m, self.eng.adam_v, self.eng.adam_count, self.eng.losses_accum), state):
            dst.copy_(src)
        if tree is not None:
            for dst, src in zip((tree._nodes_dev, tree._max_dev, tree._status), tree_state):
                dst.copy_(src)
        # the restored parameters are not what the warm-up's optimizer left in the mirror
        self.eng.invalidate_mirror()
        self.graph = g

    def run(self) -> None:
        """S steps: draw S index rows on the host stream of the sampler, stage them, replay the graph."""
        rb = self.rb
        rb._flush()
        if rb._frames.data_ptr() != self._frames_ptr:  # the frame store was re-allocated: pointers in the graph are stale
            self._frames_ptr = rb._frames.data_ptr()
            self.destroy()
            self._make_batches()
            self._capture()
        sampler = rb._sampling_distribution
        rows = sampler.draw_rows_device(self.S, self.B)
        self.block.copy_(rows, non_blocking=True)
        if self.eng.trust_mirror:
            self.eng.refresh_mirror()  # (trusted engines only: one eager launch when the bookkeeping says the mirror is stale)
        self.graph.replay()
        self.eng._mirror_made_current()  # the last step's optimizer wrote both forms of the weights
