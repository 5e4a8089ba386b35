"""Single-environment adapters with the reference's `gym.Env` surface, backed by the batched HIP path (N = 1).

`PogostickV1Env` / `BowV1Env` mirror gym_novel_gridworlds/envs/pogostick_v1_env.py and bow_v1_env.py as seen by
callers: constructor `__init__(env=None)` :26, `reset(map_size=None, items_id=None, items_quantity=None)` :86,
`step(action_id)` :230 -> `(obs, reward, done, info)` with info `{'result', 'step_cost', 'message'}` :359, the Dict
observation of `get_observation()` :214-228 whose values alias the env's attributes, the public attribute surface
 and the exceptions (ValueError for an unknown action id :236, AssertionError when items cannot be
placed :167).  Between calls the HOST attributes are the truth - exactly like the reference, callers may mutate
`env.map`, `env.inventory_items_quantity`, `env.agent_location`, ... directly (tests/keyboard_interface.py:93-100);
every step pushes them to the device, runs ONE launch of the step kernel and pulls the result back.

This is BASELINE.json config 1 ("plumbing"): it exists so the path drops in under `gym.make`; throughput comes
from `VecNovelGridworld`.
"""
import copy


import numpy as np

from . import spaces
from .spec import DIRECTION_ID, DIRECTION_STR, STEP_COSTS, EnvSpec
from .vec_env import PLACEMENT_MESSAGE, VecNovelGridworld

try:
    import gym as _gym
    _EnvBase = _gym.Env
except Exception:                                           # noqa: BLE001 - gym is optional
    _EnvBase = object

_DR, _DC = (-1, 1, 0, 0), (0, 0, -1, 1)


class _NovelGridworldEnv(_EnvBase):
    ENV_ID = None

    def __init__(self, env=None):
        self._spec = EnvSpec(self.ENV_ID, 10)
        self._vec = None
        self._vec_key = None
        self._vec_fp = None
        self._vec_cache = {}                                 # compiled-spec key -> device handle (most recently used last)
        self._seed = None
        self._episode_base = 0
        self._fp_cache = {}
        sp = self._spec
        self.env_id = sp.env_id
        self.env = env                                       # env to restore in reset (pogostick_v1_env.py:29, :89-109)
        self.map_size = sp.map_size
        self.map = np.zeros((self.map_size, self.map_size), dtype=int)
        self.agent_location = (1, 1)
        self.direction_id = dict(DIRECTION_ID)
        self.agent_facing_str = 'NORTH'
        self.agent_facing_id = self.direction_id[self.agent_facing_str]
        self.block_in_front_str = 'air'
        self.block_in_front_id = 0
        self.block_in_front_location = (0, 0)
        # the spec object's tables ARE the env's tables (novelty injection edits them in place, like the reference)
        self.items = sp.items
        self.items_id = sp.items_id
        self.unbreakable_items = sp.unbreakable_items
        self.goal_item_to_craft = sp.goal_item_to_craft
        self.items_quantity = sp.items_quantity
        self.inventory_items_quantity = {item: 0 for item in self.items}
        self.selected_item = ''
        self.entities = sp.entities
        self._avail, self._not_avail, self._avail_stale = [], [], False   # available_locations / not_available_locations (properties below)
        self.actions_id = sp.actions_id
        self.manipulation_actions_id = sp.manipulation_actions_id
        self.recipes = sp.recipes
        self.craft_actions_id = sp.craft_actions_id
        self.select_actions_id = sp.select_actions_id
        self.action_space = spaces.Discrete(len(self.actions_id))
        self.last_action = 'Forward'
        self.step_count = 0
        self.last_step_cost = 0
        self.max_items = sp.max_items
        self.observation_space = spaces.Dict({'map': spaces.Box(low=0, high=self.max_items,
                                                                shape=(self.map_size, self.map_size, 1))})
        self.last_reward = 0
        self.reward_intermediate = sp.reward_intermediate
        self.reward_done = sp.reward_done
        self.last_done = False
        self._known, self._dev_state_of = None, None         # what the device is known to hold (envs.py _push)
        self._m8 = self._i32 = self._inv_known = None        # raw bytes of the device's map / inventory row at the last pull
        self._rev_actions = self._rev_items = None           # (table identity, size, {id: name}) of actions_id / items_id

    # ------------------------------------------------------------------ backend
    def _make_backend(self, spec, seed):
        """The one place the adapter creates its one-env device handle."""
        # (prepared next episodes, refilled only after an explicit reset: reset() then copies a row instead of running the placement loop)
        return VecNovelGridworld(spec=spec, num_envs=1, seed=seed, reset_prefetch=1 << 20)

    def seed(self, seed=None):
        """The reference ignores seed() (global np.random); here it keys the device's per-episode Philox streams."""
        self._seed = None if seed is None else int(seed)
        self._close_backend()
        return [seed]

    def _sync_spec(self):
        """The spec follows the env's public attributes (callers edit / rebind them, e.g. env.map_size = 16)."""
        sp = self._spec
        sp.items, sp.items_id, sp.items_quantity, sp.entities = self.items, self.items_id, self.items_quantity, self.entities
        sp.actions_id, sp.recipes, sp.unbreakable_items = self.actions_id, self.recipes, self.unbreakable_items
        sp.goal_item_to_craft = self.goal_item_to_craft
        sp.map_size = int(self.map_size)
        sp.reward_intermediate, sp.reward_done = self.reward_intermediate, self.reward_done
        return sp

    def _tables(self):
        """What the compiled spec was built from, as the LIVE objects.  The reference reads `self.recipes`, `items_id`,
        `actions_id` ... on every step, and its users edit them in place between steps (a wrapper that changes a recipe's
        output mid-episode), so identities and sizes are not enough: every step compares this tuple with a deep copy taken
        when the handle was last checked - one `==` that walks the small tables by CONTENT in C (~0.5 us; serialising them
        for a hashable fingerprint cost 1.2 us) - plus the spec's edit counter (every novelty injection appends to it)."""
        sp = self._spec
        return (self.map_size, self.recipes, self.items_id, self.actions_id, self.items_quantity, self.entities, self.unbreakable_items,
                self.goal_item_to_craft, self.reward_done, self.reward_intermediate, len(sp.novelties), sp.fire_wall, sp.break_increase)

    def _backend(self, full_check=False):
        now = self._tables()
        if self._vec is not None and not full_check and now == self._vec_fp:
            return self._vec
        sp = self._sync_spec()      # callers may REBIND the public tables (env.items_quantity = {...}): the spec follows the env's attributes
        fp = self._vec_fp
        if self._vec is not None and fp is not None and now[1:] == fp[1:]:
            # only the map size may differ from what the current handle was built for (the reference's own loop, tests/random_action.py:51-64,
            # changes it every ten steps): the rest of the key - a dozen reprs of small tables - is the current one's
            key = (sp.map_size,) + self._vec_key[1:]
        else:
            key = (sp.map_size, tuple(sp.items_id.items()), tuple(sp.actions_id.items()), tuple(sp.items_quantity.items()),
                   tuple(sorted(sp.entities)), repr(sp.axe), repr(sp.additem), repr(sp.start_inventory), sp.reward_done,
                   sp.reward_intermediate, repr(sp.replace), repr(sp.fence), repr(sp.fence_pred), repr(sp.reset_passes), repr(sp.fire_wall), repr(sp.crate),
                   tuple(sorted(sp.unbreakable_items)), repr(sp.recipes), repr(sp.break_increase))
        if self._vec is None or key != self._vec_key:
            # a handle per compiled spec, kept: the reference's own loop (tests/random_action.py:51-64) changes map_size every
            # ten steps, and building a device handle costs milliseconds.  The episode counter travels with the env, not
            # the handle, so no (seed, episode) stream is ever replayed: `_episode_base` counts this env's resets on the host
            # (every reset1() advances the device's counter by one, whatever its outcome) and a handle that comes back into
            # use is told where the env stands.
            if self._seed is None:                          # reproducible under np.random.seed(), like the reference
                self._seed = int(np.random.randint(0, 2 ** 31 - 1))
            if self._vec is not None and hasattr(self._vec, 'sync'):
                # the handle that goes out of use may still have its resident step loop on the device (it ends by itself after 300 us without a
                # command): ended now, so that the incoming handle's launches do not queue up behind it (HIP maps streams onto a few hardware queues)
                self._vec.sync()
            vec = self._vec_cache.pop(key, None)
            if vec is None:
                import copy as _copy
                vec = self._make_backend(_copy.deepcopy(sp), self._seed)
                vec._adapter_episode = 0
                while len(self._vec_cache) >= self._VEC_CACHE:     # oldest out
                    self._vec_cache.pop(next(iter(self._vec_cache))).close()
            self._vec_cache[key] = vec                          # (re-inserted: most recently used last)
            self._vec, self._vec_key = vec, key
            self._dev_state_of = None
            if getattr(vec, '_adapter_episode', 0) != self._episode_base:
                vec.set_state(0, episode=np.array([self._episode_base], np.uint32))
                vec._adapter_episode = self._episode_base
        if now != self._vec_fp:                              # (reset() re-checks the key every time: the snapshot only when a table changed)
            snap = self._fp_cache.get(key)                   # (a handle that comes back: its tables' snapshot too, if nothing was edited meanwhile)
            if snap is None or snap != now:
                snap = self._fp_cache[key] = copy.deepcopy(now)
                while len(self._fp_cache) > 2 * self._VEC_CACHE:
                    self._fp_cache.pop(next(iter(self._fp_cache)))
            self._vec_fp = snap
        return self._vec

    _VEC_CACHE = 16

    def _close_backend(self):
        for vec in self._vec_cache.values():
            vec.close()
        self._vec_cache.clear()
        self._vec = None
        self._vec_fp = None

    def _push(self, vec):
        """host attributes -> device state.  The caller may have edited any attribute since the last pull - or nothing at all
        (the usual case in a step loop): what the device is known to hold is remembered as plain Python values and one int8
        map copy, so the unchanged case costs a handful of comparisons and no array is built."""
        sel = self.selected_item
        k = self._known
        if (self._dev_state_of is vec and k is not None and k[0] == self.agent_location and k[1] == self.agent_facing_id and k[2] == sel
                and k[3] == self.step_count and k[4] == self.inventory_items_quantity and self.map.shape == k[6]
                and self.map.tobytes() == k[5]):
            return
        ids, S, K = self.items_id, self.map_size, len(self.items_id)
        inv = np.zeros((1, K), np.int32)
        for name, q in self.inventory_items_quantity.items():
            inv[0, ids[name]] = q                            # KeyError for an unknown item name, like the reference
        m = np.ascontiguousarray(np.asarray(self.map).reshape(1, S * S), np.int8)
        vec.set_state(0, map=m, loc=np.array([self.agent_location], np.int32), facing=np.array([self.agent_facing_id], np.int32), inv=inv,
                      selected=np.array([ids[sel] if sel else 0], np.int32), step_count=np.array([self.step_count], np.int32))
        self._remember(vec)

    def _remember(self, vec):
        self._known = (tuple(self.agent_location), self.agent_facing_id, self.selected_item, self.step_count,
                       dict(self.inventory_items_quantity), self.map.tobytes(), self.map.shape)
        self._m8 = self._i32 = None                          # the next pull rebuilds its caches
        self._dev_state_of = vec

    def _pull(self, vec, st=None):
        """device state -> host attributes (the map array object is kept: observations alias it).  `st` None: the host buffers
        the last step1() / reset1() filled; else a get_state() dict."""
        if st is None:
            mb, r, c, f, ib, sel, steps = vec.last_state1()
        else:
            mb, ib = st['map'][0].tobytes(), st['inv'][0].tobytes()
            r, c, f = int(st['loc'][0][0]), int(st['loc'][0][1]), int(st['facing'][0])
            sel, steps = int(st['selected'][0]), int(st['step_count'][0])
        S = self.map_size
        # most steps change neither the map nor the inventory: compare the raw rows with the last pull's before rebuilding anything
        if mb != self._m8 or self.map.shape != (S, S) or self._known is None or self._dev_state_of is not vec:
            if self.map.shape != (S, S):
                self.map = np.zeros((S, S), dtype=int)
            self.map[...] = np.frombuffer(mb, np.int8).reshape(S, S)
            self._m8 = mb
            m64 = self.map.tobytes()
        else:
            m64 = self._known[5]
        self.agent_location = (r, c)
        self.agent_facing_id = f
        self.agent_facing_str = DIRECTION_STR[f]
        names = self._spec.item_names
        inv = self.inventory_items_quantity
        if ib != self._i32 or self._inv_known != inv:
            for name, q in zip(names, np.frombuffer(ib, np.int32).tolist()):
                inv[name] = q
            self._i32, self._inv_known = ib, dict(inv)
        self.selected_item = names[sel] if sel else ''
        self.step_count = steps
        self._known = (self.agent_location, f, self.selected_item, steps, self._inv_known, m64, self.map.shape)
        self._dev_state_of = vec

    # ------------------------------------------------------------------ reference API
    def reset(self, map_size=None, items_id=None, items_quantity=None):
        if self.env is not None:                             # restore branch, pogostick_v1_env.py:89-109
            print("RESTORING " + self.env_id + " ...")
            src = self.env
            self.map_size = copy.deepcopy(src.map_size)
            self.map = copy.deepcopy(src.map)
            self.items_id.clear(); self.items_id.update(copy.deepcopy(src.items_id))
            self.items_quantity.clear(); self.items_quantity.update(copy.deepcopy(src.items_quantity))
            self.inventory_items_quantity = copy.deepcopy(src.inventory_items_quantity)
            self.available_locations = copy.deepcopy(src.available_locations)
            self.not_available_locations = copy.deepcopy(src.not_available_locations)
            self.last_action = copy.deepcopy(src.last_action)
            self.step_count = copy.deepcopy(src.step_count)
            self.last_reward = copy.deepcopy(src.last_reward)
            self.last_done = False
            self.agent_location = copy.deepcopy(src.agent_location)
            self.set_agent_facing(copy.deepcopy(src.agent_facing_str))
            obs = self.get_observation()
            self.update_block_in_front()
            return obs
        if map_size is not None:
            self.map_size = map_size
        if items_id is not None:
            self.items_id.clear(); self.items_id.update(items_id)
        if items_quantity is not None:
            self.items_quantity.clear(); self.items_quantity.update(items_quantity)
        self.inventory_items_quantity = {item: 0 for item in self.items}
        self.selected_item = ''
        self._avail, self._not_avail, self._avail_stale = [], [], True      # rebuilt from the new map when somebody looks (properties below)
        self.last_action = 'Forward'
        self.step_count = 0
        self.last_step_cost = 0
        self.last_reward = 0
        self.last_done = False
        vec = self._backend(full_check=True)
        self._episode_base += 1                              # (the device's episode counter advances with the call, whatever its outcome)
        vec._adapter_episode = self._episode_base
        vec.reset1()                                         # one C-ABI call that also brings the state back; AssertionError(PLACEMENT_MESSAGE) when items do not fit
        self._pull(vec)
        obs = self.get_observation()
        self.update_block_in_front()
        return obs

    def step(self, action_id):
        # ValueError("<id> is not in list") for an unknown action id, before anything changes (:236)
        rev = self._rev_actions
        if rev is None or rev[0] is not self.actions_id or rev[1] != len(self.actions_id):
            rev = self._rev_actions = (self.actions_id, len(self.actions_id), {v: k for k, v in reversed(list(self.actions_id.items()))})
        name = rev[2].get(action_id)
        if name is None or self.actions_id.get(name) != action_id:      # (table edited in place, or an unknown id: the reference's own lookup)
            self._rev_actions = None
            name = list(self.actions_id.keys())[list(self.actions_id.values()).index(action_id)]
        self.last_action = name
        vec = self._backend()
        self._push(vec)
        reward, done, result, cost_code, msg_code, msg_arg = vec.step1(action_id)   # one C-ABI call, scalars back (no per-step arrays)
        self._pull(vec)                                      # the call already brought the state back
        obs = self.get_observation()
        self.update_block_in_front()
        step_cost = STEP_COSTS[cost_code]
        out_info = {'result': result, 'step_cost': step_cost, 'message': self._spec.format_message(int(action_id), msg_code, msg_arg)}
        self.last_step_cost = step_cost
        self.last_reward = reward
        self.last_done = done
        return obs, reward, done, out_info

    def get_observation(self):
        assert not self.max_items < len(self.items), "Cannot have more than " + str(self.max_items) + " items"
        return {'map': self.map, 'agent_location': self.agent_location, 'agent_facing_id': self.agent_facing_id,
                'inventory_items_quantity': self.inventory_items_quantity}

    def set_agent_location(self, r, c):
        self.agent_location = (r, c)

    def set_agent_facing(self, direction_str):
        self.agent_facing_str = direction_str
        self.agent_facing_id = self.direction_id[self.agent_facing_str]

    def set_lasts(self, lasts):
        self.last_action = lasts['last_action']
        self.step_count = lasts['step_count']
        self.last_step_cost = lasts['last_step_cost']
        self.last_reward = lasts['last_reward']
        self.last_done = lasts['last_done']

    def set_items_id(self, items):
        from .spec import set_items_id
        return set_items_id(items)

    def update_block_in_front(self):                         # pogostick_v1_env.py:369-389
        r, c = self.agent_location
        f = self.agent_facing_id
        self.block_in_front_location = (r + _DR[f], c + _DC[f])
        self.block_in_front_id = int(self.map[self.block_in_front_location[0]][self.block_in_front_location[1]])
        if self.block_in_front_id == 0:
            self.block_in_front_str = 'air'
        else:
            rev = self._rev_items
            if rev is None or rev[0] is not self.items_id or rev[1] != len(self.items_id):
                rev = self._rev_items = (self.items_id, len(self.items_id), {v: k for k, v in reversed(list(self.items_id.items()))})
            name = rev[2].get(self.block_in_front_id)
            if name is None or self.items_id.get(name) != self.block_in_front_id:
                self._rev_items = None
                name = list(self.items_id.keys())[list(self.items_id.values()).index(self.block_in_front_id)]
            self.block_in_front_str = name

    def is_block_in_front_next_to(self, item):               # :391-411
        self.update_block_in_front()
        r, c = self.block_in_front_location
        for d in range(4):
            rr, cc = r + _DR[d], c + _DC[d]
            if 0 <= rr <= self.map_size - 1 and 0 <= cc <= self.map_size - 1 and self.map[rr][cc] == self.items_id[item]:
                return True
        return False

    def add_new_items(self, new_items_quantity):             # :495-501
        for item in new_items_quantity:
            self._spec.add_new_item(item)
            self.items_quantity.update({item: new_items_quantity[item]})
        self.reset()

    # ---- public map / table editing helpers of the reference env (host attributes are the truth between calls; the next
    #      step() pushes them to the device).  reset() and step() run placement and crafting in the kernels; add_item_to_map and
    #      craft below are the same rules as public host-side methods, for callers that use them directly.
    def remap_action(self, actions_id, start_action_id):
        """Shuffle action names with the global numpy stream until the table changes (:476-493)."""
        from .novelty import _remap_action
        return _remap_action(actions_id, start_action_id)

    def add_fence_around(self, item_location, fence_name):
        """Every free 8-neighbour of `item_location` except the agent cell becomes `fence_name` (:524-536)."""
        r, c = item_location
        for rr in (r - 1, r, r + 1):
            for cc in (c - 1, c, c + 1):
                if self.map[rr][cc] == 0 and (rr, cc) != self.agent_location:
                    self.map[rr][cc] = self.items_id[fence_name]

    def block_items(self, item_to_block, item_to_block_from):
        """Put `item_to_block_from` on the free in-bounds 4-neighbours of every `item_to_block` (:503-522)."""
        rows, cols = np.where(self.map == self.items_id[item_to_block])
        for r, c in zip(rows, cols):
            for rr, cc in ((r - 1, c), (r + 1, c), (r, c - 1), (r, c + 1)):
                if 0 <= rr <= self.map_size - 1 and 0 <= cc <= self.map_size - 1 and self.map[rr][cc] == 0 \
                        and (rr, cc) != self.agent_location:
                    self.map[rr][cc] = self.items_id[item_to_block_from]

    def grab_entities(self, location=None):
        """Entities in the 3x3 around `location` (default: the agent) go to the inventory (:538-554)."""
        r, c = self.agent_location if location is None else location
        names = {v: k for k, v in self.items_id.items()}
        for rr in (r - 1, r, r + 1):
            for cc in (c - 1, c, c + 1):
                ent = int(self.map[rr][cc])
                if ent != 0 and names[ent] in self.entities:
                    self.map[rr][cc] = 0
                    self.inventory_items_quantity[names[ent]] += 1

    # `available_locations` / `not_available_locations` (:136-138, :181).  In the reference they are what reset()'s placement loop
    # left behind: the interior candidates [2, S-3]^2 it never drew (+ the agent's cell), and the ones it drew and popped.  reset()
    # here runs that loop in the kernels on a per-episode Philox stream, so WHICH blocked candidates were drawn and discarded is not
    # known on the host; what is known is the invariant that matters to a later add_item_to_map(): every cell that can still take an
    # item (itself and its four neighbours air) was never drawn - a drawn one would hold an item - so it is still a candidate.  The
    # lists are therefore rebuilt from the map when first looked at after a reset(): candidates = interior cells that hold no item
    # (row-major; the agent's cell among them, as in the reference), popped = interior cells that hold one.  A later
    # add_item_to_map() then draws the same distribution over the cells that can take the item, and "Cannot place items" fires under
    # the same condition (no such cell left) - after more discarded draws than in the reference, whose list is shorter.
    def _rebuild_locations(self):
        self._avail_stale = False
        inner = range(2, self.map_size - 2)
        grid = self.map
        self._avail = [(r, c) for r in inner for c in inner if not grid[r][c]]
        self._not_avail = [(r, c) for r in inner for c in inner if grid[r][c]]

    @property
    def available_locations(self):
        if self._avail_stale:
            self._rebuild_locations()
        return self._avail

    @available_locations.setter
    def available_locations(self, value):
        if self._avail_stale:
            self._rebuild_locations()
        self._avail = value

    @property
    def not_available_locations(self):
        if self._avail_stale:
            self._rebuild_locations()
        return self._not_avail

    @not_available_locations.setter
    def not_available_locations(self, value):
        if self._avail_stale:
            self._rebuild_locations()
        self._not_avail = value

    def add_item_to_map(self, item, num_items):
        """Public form of the placement loop (:159-181) on the host attributes, drawing from the global numpy stream like the
        reference: a random remaining candidate; the agent's cell is dropped; a cell whose 4-neighbourhood is all air takes the
        item; every drawn candidate leaves the list.  (reset() itself runs the same loop in the kernels, on its per-episode
        Philox stream; the candidate list this call works on is `available_locations` - see above for what it holds after a reset().)"""
        candidates, grid, placed = self.available_locations, self.map, 0
        item_id = self.items_id[item]
        while placed != num_items:
            assert candidates, PLACEMENT_MESSAGE
            pick = int(np.random.choice(len(candidates), size=1)[0])
            r, c = spot = candidates.pop(pick)
            if spot == tuple(self.agent_location):
                continue
            if not (grid[r][c] or grid[r - 1][c] or grid[r + 1][c] or grid[r][c - 1] or grid[r][c + 1]):
                grid[r][c] = item_id
                placed += 1
            self.not_available_locations.append(spot)

    def craft(self, item_to_craft):
        """Public form of the craft rule (:413-474) on the host attributes: (reward, result, step_cost, message).  The costs and
        the reward are the compiled spec's per-recipe table entries - the ones the kernels use for the Craft_* actions."""
        spec = self._sync_spec()
        needs = self.recipes[item_to_craft]['input']
        have = self.inventory_items_quantity
        cost_missing, cost_no_table, cost_ok = spec.craft_costs.get(item_to_craft, (0, 0, 0))
        short = [name for name, q in needs.items() if have.get(name, -1) < q]
        if short:
            return -1, False, cost_missing, "Missing items: " + ", ".join("%s %s" % (needs[name], name) for name in short)
        if len(needs) > 1:
            self.update_block_in_front()
            if self.block_in_front_str != 'crafting_table':
                return -1, False, cost_no_table, 'Need to be in front of crafting_table'
        for name, q in needs.items():
            have[name] -= q
        have[item_to_craft] += self.recipes[item_to_craft]['output'][item_to_craft]
        return spec.recipe_rewards.get(item_to_craft, spec.craft_reward), True, cost_ok, 'Crafted ' + item_to_craft

    def render(self, mode='human', title=None):
        raise NotImplementedError("rendering is outside the batched hot path (SURVEY.md §2 row 12)")

    def close(self):
        self._close_backend()


class PogostickV1Env(_NovelGridworldEnv):
    """Goal: craft 1 pogo_stick (pogostick_v1_env.py:17-24)."""
    ENV_ID = 'NovelGridworld-Pogostick-v1'


class BowV1Env(_NovelGridworldEnv):
    """Goal: craft 1 bow (bow_v1_env.py:17-24)."""
    ENV_ID = 'NovelGridworld-Bow-v1'


class PogostickV0Env(_NovelGridworldEnv):
    """pogostick_v0_env.py: starts with sticks / planks on the map and a tree_tap already next to a tree_log."""
    ENV_ID = 'NovelGridworld-Pogostick-v0'


class BowV0Env(_NovelGridworldEnv):
    """bow_v0_env.py: sticks and strings lie on the map."""
    ENV_ID = 'NovelGridworld-Bow-v0'


ENTRY_POINTS = {'NovelGridworld-Pogostick-v1': PogostickV1Env, 'NovelGridworld-Bow-v1': BowV1Env,
                'NovelGridworld-Pogostick-v0': PogostickV0Env, 'NovelGridworld-Bow-v0': BowV0Env}


def make(env_id, **kwargs):
    """`gym.make(id)` equivalent that works without gym (entry points as in gym_novel_gridworlds/__init__.py:47-60)."""
    return ENTRY_POINTS[env_id](**kwargs)
