"""TEST INFRASTRUCTURE ONLY -- drives the reference's own step body without run().

`OpenDriftSimulation.run()` builds a real xarray Dataset (xarray is not installed
here), so this restates the *orchestration only* of
opendrift/models/basemodel/__init__.py:1895-2284 -- every arithmetic call goes to
the reference's own methods (get_environment, update, horizontal_diffusion,
interact_with_coastline, ...) imported through oracle/refshim.py.  The live
float64 state o.elements.{lon,lat,z} is recorded after every step (never the
float32 history buffer, basemodel/__init__.py:2094-2104).
"""
from datetime import timedelta
from types import SimpleNamespace

import numpy as np


class RecordingRandom:
    """Context manager: records every np.random.random / normal / uniform / rand draw in call order."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = (np.random.random, np.random.normal, np.random.uniform, np.random.rand)
        rec = self

        def rand(*shape):
            r = rec._orig[3](*shape)
            rec.draws.append(('rand', np.array(r, copy=True)))
            return r

        def random(size=None):
            r = rec._orig[0](size)
            rec.draws.append(('random', np.array(r, copy=True)))
            return r

        def normal(loc=0.0, scale=1.0, size=None):
            r = rec._orig[1](loc, scale, size)
            rec.draws.append(('normal', np.array(r, copy=True), loc, scale))
            return r

        def uniform(low=0.0, high=1.0, size=None):
            r = rec._orig[2](low, high, size)
            rec.draws.append(('uniform', np.array(r, copy=True), low, high))
            return r

        np.random.random, np.random.normal, np.random.uniform, np.random.rand = random, normal, uniform, rand
        return self

    def __exit__(self, *a):
        np.random.random, np.random.normal, np.random.uniform, np.random.rand = self._orig


class RefStepper:
    def __init__(self, o, time_step, steps=1000000):
        from opendrift.models.basemodel import Mode, evaluate_conditional
        self.o = o
        if not isinstance(time_step, timedelta):
            time_step = timedelta(seconds=time_step)
        # --- run() preamble, basemodel/__init__.py:1895-1924: conditionals
        for vn, var in o.required_variables.copy().items():
            if 'skip_if' in var and evaluate_conditional(*var['skip_if'], o) is True:
                o.required_variables.pop(vn)
        for en, prop in o.elements.variables.copy().items():
            if 'store_previous_if' in prop:
                if evaluate_conditional(*prop['store_previous_if'], o) is True:
                    o.elements.variables[en]['store_previous'] = True
                del o.elements.variables[en]['store_previous_if']
        for en, var in o.required_variables.copy().items():
            if 'store_previous_if' in var:
                if evaluate_conditional(*var['store_previous_if'], o) is True:
                    o.required_variables[en]['store_previous'] = True
                del o.required_variables[en]['store_previous_if']
        o.time_step = time_step
        o.time_step_output = time_step
        o.expected_steps_calculation = steps
        o.time = o.start_time
        # --- readers prepared for the simulation extent, :2018-2046
        max_distance = o.get_config('drift:max_speed') * min(steps, 1000) * abs(time_step.total_seconds())
        dlat = max_distance / 111000.
        dlon = dlat / np.cos(np.radians(np.mean(o.elements_scheduled.lat)))
        ext = np.array([max(-360, o.elements_scheduled.lon.min() - dlon),
                        max(-89, o.elements_scheduled.lat.min() - dlat),
                        min(360, o.elements_scheduled.lon.max() + dlon),
                        min(89, o.elements_scheduled.lat.max() + dlat)])
        o.simulation_extent = ext
        o.env.finalize(simulation_extent=ext, start=o.start_time,
                       end=o.start_time + min(steps, 100000) * time_step)
        o.mode = Mode.Run
        # previous positions (xarray Dataset in the reference, :2163-2166): plain arrays by ID
        n_total = len(o.elements_scheduled)
        store_prev = o.elements.variables['lon'].get('store_previous', False)
        o._elements_previous = SimpleNamespace(
            lon=np.array(o.elements_scheduled.lon, dtype=np.float64),
            lat=np.array(o.elements_scheduled.lat, dtype=np.float64),
            __contains__=lambda k: True) if store_prev else None
        self._store_prev = store_prev
        o._environment_previous = None
        o.validity_domain = None
        o.steps_calculation = 0
        o.prepare_run()
        self.n_total = n_total

    def _release(self):
        o = self.o
        if self._store_prev:
            prev = o._elements_previous
            o._elements_previous = None  # release_elements tests `'lon' in self._elements_previous`
            sched_ID = np.array(o.elements_scheduled.ID, copy=True)
            sched_lon = np.array(o.elements_scheduled.lon, copy=True)
            sched_lat = np.array(o.elements_scheduled.lat, copy=True)
            o.release_elements()
            o._elements_previous = prev
            if o.newly_seeded_IDs is not None and len(o.newly_seeded_IDs):
                sel = np.isin(sched_ID, o.newly_seeded_IDs)
                prev.lon[sched_ID[sel]] = sched_lon[sel]
                prev.lat[sched_ID[sel]] = sched_lat[sel]
        else:
            o.release_elements()

    def step(self):
        """One pass of the loop body, basemodel/__init__.py:2193-2284 (no state_to_buffer)."""
        o = self.o
        self._release()
        o.environment, o.environment_profiles, missing = o.env.get_environment(
            list(o.required_variables), o.time, o.elements.lon, o.elements.lat, o.elements.z,
            o.required_profiles, o.profiles_depth, element_ID=o.elements.ID)
        o.calculate_missing_environment_variables()
        o.report_missing_variables(missing)
        o.deactivate_outside()
        o.interact_with_coastline()
        o.interact_with_seafloor()
        o.increase_age_and_retire()
        o.remove_deactivated_elements()
        if self._store_prev:  # update_previous_state, :642-668, without xarray
            o._elements_previous.lon[o.elements.ID] = o.elements.lon
            o._elements_previous.lat[o.elements.ID] = o.elements.lat
        if o.num_elements_active() > 0:
            o.update()
        o.horizontal_diffusion()
        o.time = o.time + o.time_step
        o.steps_calculation += 1

    def state(self):
        o = self.o
        e, d = o.elements, o.elements_deactivated
        n = self.n_total
        lon = np.full(n, np.nan)
        lat = np.full(n, np.nan)
        z = np.full(n, np.nan)
        status = np.full(n, -1, np.int32)
        for arr in (e, d):
            if len(arr) == 0:
                continue
            ID = np.atleast_1d(arr.ID).astype(int)
            lon[ID] = arr.lon
            lat[ID] = arr.lat
            z[ID] = np.atleast_1d(arr.z) * np.ones(len(ID))
            status[ID] = np.atleast_1d(arr.status) * np.ones(len(ID))
        return lon, lat, z, status
