"""Wall-clock timer buckets and signature introspection.

Counterparts of the reference's `misc_utilities.Timer` (misc_utilities.py:22-60; same
bucket names: step, cont, wind, temp, moist, diag) and `function_input_fields`
(misc_utilities.py:63-73).
"""
import time
from inspect import signature


class Timer:

    def __init__(self, sync=None):
        self.timings = {}
        self.flags = {}
        self.sync = sync          # callable that drains the device queue (i_sync_context = 1)

    def start(self, timer_key):
        self.timings.setdefault(timer_key, 0.)
        self.flags[timer_key] = time.time()

    def stop(self, timer_key):
        if self.flags.get(timer_key) is None:
            raise ValueError('No time measurement in progress for timer ' + str(timer_key) + '.')
        if self.sync is not None:
            self.sync()
        self.timings[timer_key] += time.time() - self.flags[timer_key]
        self.flags[timer_key] = None

    def print_report(self):
        total = self.timings.get('total', sum(self.timings.values()))
        print('took %.2f min.' % (total / 60))
        print('Detailed computing times:')
        for key, value in self.timings.items():
            print('%s\t%.0f\t%%\t%.1f \tsec' % (key, 100 * value / total if total else 0., value))


def function_input_fields(function):
    """Model-field names a factory method takes (its signature minus self/GR/GRF)."""
    return [n for n in signature(function).parameters if n not in ('self', 'GR', 'GRF')]
