"""LidarInFront observation: host-side tables (reference gym_novel_gridworlds/observation_wrappers.py:10-80).

The reference shoots `num_beams` rays with float arithmetic: `x_ratio, y_ratio = np.round(np.cos(angle), 2),
np.round(np.sin(angle), 2)` and `r_obj = r + np.round(beam_range * x_ratio)` (:49-55).  Those are pure functions of
(facing, beam, range), so they are evaluated HERE with the same numpy calls and handed to the kernel as integer
offset tables - the device never touches a float."""
import ctypes as C
import math

import numpy as np

from .spec import MAX_ITEMS

MAX_BEAMS, MAX_RANGE = 16, 64


class NgwLidarCfg(C.Structure):
    """ctypes mirror of `struct ngw_lidar_cfg` (include/ngw.h)."""
    _fields_ = [('num_beams', C.c_int32), ('max_range', C.c_int32), ('n_chan', C.c_int32), ('n_inv', C.c_int32),
                ('chan_of_item', C.c_uint8 * MAX_ITEMS), ('inv_item', C.c_uint8 * MAX_ITEMS),
                ('dr', ((C.c_int8 * MAX_RANGE) * MAX_BEAMS) * 4), ('dc', ((C.c_int8 * MAX_RANGE) * MAX_BEAMS) * 4)]


class LidarConfig:
    """What `LidarInFront.__init__` fixes at wrap time (:16-30) + what `observation()` reads at call time (:74-75)."""

    def __init__(self, spec, num_beams=8):
        self.num_beams = int(num_beams)
        # :21-24 lidar items = all items except air and the goal item, ids 1.. in alphabetical order (no 'air' -> ids start at 1)
        lidar_items = set(spec.items_id.keys()) - {'air', spec.goal_item_to_craft}
        self.lidar_items_id = {item: i + 1 for i, item in enumerate(sorted(lidar_items))}
        self.max_beam_range = int(math.sqrt(2 * (spec.map_size - 2) ** 2))      # :25
        if not (1 <= self.num_beams <= MAX_BEAMS) or self.max_beam_range > MAX_RANGE:
            raise ValueError("num_beams must be in [1, %d] and the beam range <= %d" % (MAX_BEAMS, MAX_RANGE))

    def obs_len(self, spec):
        return self.num_beams * len(self.lidar_items_id) + len(self.inventory_order(spec))

    @staticmethod
    def inventory_order(spec):
        """[inventory[item] for item in sorted(inventory) if item not in unbreakable_items] (:74-75), at call time."""
        return [item for item in sorted(spec.items) if item not in spec.unbreakable_items]

    def compile(self, spec):
        """Flatten against the CURRENT spec (items added by a later novelty are not lidar items but are in the inventory)."""
        c = NgwLidarCfg()
        c.num_beams, c.max_range, c.n_chan = self.num_beams, self.max_beam_range, len(self.lidar_items_id)
        for item, ch in self.lidar_items_id.items():
            c.chan_of_item[spec.items_id[item]] = ch
        order = self.inventory_order(spec)
        c.n_inv = len(order)
        for j, item in enumerate(order):
            c.inv_item[j] = spec.items_id[item]
        direction_radian = [np.pi, 0, 3 * np.pi / 2, np.pi / 2]                # NORTH SOUTH WEST EAST (:38)
        for f in range(4):
            angles = np.linspace(direction_radian[f] - np.pi, direction_radian[f] + np.pi, self.num_beams + 1)[:-1]   # :41-43
            for b, angle in enumerate(angles):
                x_ratio, y_ratio = np.round(np.cos(angle), 2), np.round((np.sin(angle)), 2)      # :49
                for k in range(1, self.max_beam_range + 1):
                    c.dr[f][b][k - 1] = int(np.round(k * x_ratio))                               # :54
                    c.dc[f][b][k - 1] = int(np.round(k * y_ratio))                               # :55
        return c
