"""CPU-oracle instantiation of the synthetic projects (groove_amd/projects.py `plan`).

TEST INFRASTRUCTURE ONLY (see oracle_dsp.hpp): used by tests/, and by bench.py for its
`cpu_baseline` leg and for the sampled parity figure it prints beside every timed workload —
never by the product package.  The plan is data (parameter arrays, note events per block); the
arithmetic here is the oracle's.
"""
import numpy as np

from groove_amd import abi_types as T
from groove_amd import projects as PJ
from . import oracle as O

FRAMES = T.BLOCK_FRAMES


class OracleProject:
    """The voices `sel` of a workload on the f64 scalar oracle; step() returns one block of the bus."""

    def __init__(self, workload, sel, grouped=True, bank_scale=1.0, lib_=None):
        self.period = PJ.WORKLOADS[workload]["blocks"]
        self.block_index = 0
        self.n = int(len(sel))
        self.banks = []
        for spec in PJ.plan(workload, sel, grouped, bank_scale):
            if spec["kind"] == "welsh":
                bank = O.Bank.welsh(spec["params"], lib_=lib_)
            elif spec["kind"] == "fm":
                bank = O.Bank.fm(spec["params"], lib_=lib_)
            else:
                bank = O.Bank.sampler(spec["pcm"], spec["descs"], spec["params"], lib_=lib_)
            fx = [O.Fx(k, p) for k, p in spec["fx"]]
            self.banks.append((bank, fx, spec["events"]))

    def step(self, frames=FRAMES, threads=1):
        """bus[frames][2] (f64) of the next block: events, render, effect chains, mix."""
        b = self.block_index % self.period
        self.block_index += 1
        bus = np.zeros((frames, 2), dtype=np.float64)
        for bank, fx, events in self.banks:
            ev = events.get(b)
            if ev is not None:
                bank.note_events(ev)
            if fx:
                blk = bank.render(frames)
                for e in fx:
                    e.process(blk)
                O.mix(blk, bus)
            else:
                bus += bank.render_bus(frames, threads=threads)
        return bus

    def run_mt(self, blocks, threads, frames=FRAMES):
        """`blocks` blocks of the timeline on `threads` persistent worker threads per stretch: the blocks between two note-event
        blocks are ONE spawn / join of the workers (oracle_bank_render_bus_blocks_mt), every thread keeping its voices throughout.
        Instruments without effect chains only (bench.py's all-cores baseline).  Returns (bus, number of spawn / join stretches)."""
        assert not any(fx for _, fx, _ in self.banks)
        out, stretches = [], 0
        done = 0
        while done < blocks:
            b = self.block_index % self.period
            for bank, _, events in self.banks:
                ev = events.get(b)
                if ev is not None:
                    bank.note_events(ev)
            run = 1   # ... up to the next block that carries events (or the end)
            while done + run < blocks and not any(events.get((self.block_index + run) % self.period) is not None for _, _, events in self.banks):
                run += 1
            bus = np.zeros((run * frames, 2), dtype=np.float64)
            for bank, _, _ in self.banks:
                bus += bank.render_bus_blocks(frames, run, threads)
            out.append(bus)
            self.block_index += run
            done += run
            stretches += 1
        return np.concatenate(out, axis=0), stretches

    def render(self, blocks, frames=FRAMES):
        return np.concatenate([self.step(frames) for _ in range(blocks)], axis=0)
