"""The parameters object of the reference API (optic/utils.py:29-146): a plain
attribute bag.  The propagation functions only use getattr / attribute assignment,
so any object with settable attributes (including the reference's own
``optic.utils.parameters``) works as well."""
import copy

import numpy as np

_SI = {-12: "p", -9: "n", -6: "µ", -3: "m", 0: "", 3: "k", 6: "M", 9: "G", 12: "T", 15: "P"}


class parameters:
    """Struct-like container of simulation parameters."""

    def _items(self):
        return vars(self).items()

    @staticmethod
    def _is_number(v):
        return isinstance(v, (int, float)) and not isinstance(v, bool)

    def to_engineering_notation(self, value):
        """1.0e4 <= |x| or 0 < |x| < 1e-4  ->  'mantissa prefix' with a power-of-1000 SI prefix."""
        if self._is_number(value) and (abs(value) >= 10000 or 0 < abs(value) < 0.0001):
            e3 = (int(np.floor(np.log10(abs(value)))) // 3) * 3
            return f"{value / 10 ** e3:.1f} {_SI.get(e3, '')}"
        return value

    def view(self):
        for name, value in self._items():
            big = self._is_number(value) and value > 10000
            print(f"{name}: {value:.2e}" if big else f"{name}: {value}")

    def _rows(self):
        for name, value in self._items():
            yield name, ("Array" if isinstance(value, (list, tuple, np.ndarray))
                         else self.to_engineering_notation(value))

    def table(self):
        lines = ["| Parameter Name | Value |", "|----------------|-----------------|"]
        lines += [f"| {n} | {v} |" for n, v in self._rows()]
        return print("\n".join(lines) + "\n")

    def latex_table(self):
        body = "".join(f"{n} & {v} \\\\\n\\hline\n" for n, v in self._rows())
        return print("\\begin{tabular}{|c|c|}\n\\hline\nParameter Name & Value \\\\\n\\hline\n"
                     + body + "\\end{tabular}")

    def copy(self):
        return copy.deepcopy(self)
