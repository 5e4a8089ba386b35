"""Spec compiler: environment tables -> flat integer look-up tables (`ngw_spec`, include/ngw.h).

Host-side restatement, as DATA, of the static tables the reference builds in its constructors:

* NovelGridworld-Pogostick-v1: gym_novel_gridworlds/envs/pogostick_v1_env.py:26-84
  (items :39, set_items_id :200-212, items_quantity :44, manipulation actions :53, recipes :56-59,
  Craft_* for sorted recipes :61-62, Select_* for sorted(items ^ unbreakable) :65-66, rewards :81-82)
* NovelGridworld-Bow-v1: gym_novel_gridworlds/envs/bow_v1_env.py:26-82
* step costs / rewards per branch: pogostick_v1_env.py:244-347, :413-474; bow_v1_env.py:293-304, :386-441
* novelty table edits: novelty_wrappers.py:125-134 (AxeMedium), :16-27 (AxeEasy), :996-1011 (AddItem)

The kernels never see names, floats or strings: step costs travel as codes into `STEP_COSTS`
(the reference mixes Python float and int costs; 27.906975 is not float32-representable) and
messages as (code, arg) pairs formatted here by `format_message`.
"""
import copy
import ctypes as C

MAX_ITEMS, MAX_ACTIONS, MAX_RECIPES, MAX_RECIPE_INPUTS, MAX_START, MAX_INV_START = 24, 48, 8, 4, 8, 4
MAX_MAP_SIZE = 64
MAX_PASSES = 4
ABI_VERSION = 3

# step_cost values with their Python types (SURVEY.md §8(a) "Distinct step_cost values")
STEP_COSTS = [0, 24.0, 27.906975, 120.0, 300.0, 360.0, 480.0, 720.0, 840.0, 1200.0, 1800.0, 2400.0, 3600.0,
              7200.0, 8400.0, 5000, 50000, 900.0, 3600.0 * 1.2, 27.906975 * 2, 600.0, 6000.0]


def cost_code(value):
    for i, v in enumerate(STEP_COSTS):
        if type(v) is type(value) and v == value:
            return i
    raise KeyError("step cost %r has no code" % (value,))


ACT_FORWARD, ACT_LEFT, ACT_RIGHT, ACT_BREAK, ACT_PLACE, ACT_EXTRACT, ACT_CRAFT, ACT_SELECT, ACT_CHOP, ACT_JUMP = range(10)
(MSG_NONE, MSG_BLOCK_IN_PATH, MSG_CANNOT_BREAK, MSG_PLACED, MSG_ALREADY_EXISTS, MSG_NOT_IN_INVENTORY,
 MSG_EXTRACT_NO_SRC, MSG_EXTRACT_NOT_NEAR, MSG_MISSING_ITEMS, MSG_NEED_TABLE, MSG_CRAFTED, MSG_NEED_AXE,
 MSG_CANNOT_CHOP, MSG_FENCE_RESTRICTION, MSG_FIRE_WALL) = range(15)

F_INVALID_ACTION, F_PLACEMENT = 1, 2

DIRECTION_ID = {'NORTH': 0, 'SOUTH': 1, 'WEST': 2, 'EAST': 3}     # pogostick_v1_env.py:33
DIRECTION_STR = ['NORTH', 'SOUTH', 'WEST', 'EAST']


class NgwSpec(C.Structure):
    """ctypes mirror of `struct ngw_spec` (include/ngw.h) - field order and types must match."""
    _fields_ = [
        ('abi_version', C.c_int32), ('map_size', C.c_int32), ('n_items', C.c_int32), ('n_actions', C.c_int32),
        ('n_recipes', C.c_int32), ('reward_step', C.c_int32), ('reward_done', C.c_int32),
        ('act_kind', C.c_uint8 * MAX_ACTIONS), ('act_arg', C.c_uint8 * MAX_ACTIONS),
        ('breakable', C.c_uint8 * MAX_ITEMS), ('entity', C.c_uint8 * MAX_ITEMS),
        ('break_reward', C.c_int8 * MAX_ITEMS), ('break_qty', C.c_uint8 * MAX_ITEMS),
        ('wall_item', C.c_uint8), ('table_item', C.c_uint8), ('goal_item', C.c_uint8), ('n_entities', C.c_uint8),
        ('recipe_in', (C.c_uint8 * MAX_ITEMS) * MAX_RECIPES),
        ('recipe_n_in', C.c_uint8 * MAX_RECIPES),
        ('recipe_in_item', (C.c_uint8 * MAX_RECIPE_INPUTS) * MAX_RECIPES),
        ('recipe_out_item', C.c_uint8 * MAX_RECIPES), ('recipe_out_qty', C.c_uint8 * MAX_RECIPES),
        ('recipe_needs_table', C.c_uint8 * MAX_RECIPES),
        ('cost_missing', C.c_uint8 * MAX_RECIPES), ('cost_no_table', C.c_uint8 * MAX_RECIPES),
        ('cost_ok', C.c_uint8 * MAX_RECIPES), ('recipe_reward', C.c_int8 * MAX_RECIPES),
        ('cost_forward', C.c_uint8), ('cost_turn', C.c_uint8), ('cost_break', C.c_uint8),
        ('cost_place', C.c_uint8), ('cost_extract', C.c_uint8), ('cost_select', C.c_uint8),
        ('cost_chop', C.c_uint8), ('cost_jump', C.c_uint8), ('chop_reward', C.c_int8), ('_pad3', C.c_uint8),
        ('place_item', C.c_uint8), ('place_near', C.c_uint8), ('place_reward', C.c_int8),
        ('ext_src', C.c_uint8), ('ext_near', C.c_uint8), ('ext_out', C.c_uint8), ('ext_qty', C.c_uint8),
        ('ext_consume', C.c_uint8), ('ext_cost_ok', C.c_uint8), ('ext_reward', C.c_int8),
        ('axe_item', C.c_uint8), ('axe_cost', C.c_uint8), ('axe_qty', C.c_uint8), ('axe_reward', C.c_int8),
        ('axe_required', C.c_uint8), ('_pad2', C.c_uint8 * 3),
        ('n_start', C.c_uint8), ('start_item', C.c_uint8 * MAX_START), ('start_qty', C.c_uint8 * MAX_START),
        ('tap_item', C.c_uint8), ('tap_near', C.c_uint8),
        ('n_inv_start', C.c_uint8), ('inv_start_item', C.c_uint8 * MAX_INV_START), ('inv_start_qty', C.c_uint8 * MAX_INV_START),
        ('n_passes', C.c_uint8), ('pass_kind', C.c_uint8 * MAX_PASSES), ('pass_item', C.c_uint8 * MAX_PASSES),
        ('pass_from', C.c_uint8 * MAX_PASSES), ('pass_pct_lo', C.c_uint8 * MAX_PASSES), ('pass_pct_hi', C.c_uint8 * MAX_PASSES),
        ('fence_item', C.c_uint8), ('fence_mode', C.c_uint8),
        ('fire_item', C.c_uint8), ('fire_reward', C.c_int8),
        ('crate_item', C.c_uint8), ('crate_add', C.c_uint8 * MAX_ITEMS),
        ('ext_flags', C.c_uint8), ('fire_skip_recipe', C.c_uint8),
    ]


# ----------------------------------------------------------------------------- environment tables
_ENV_DEFS = {
    'NovelGridworld-Pogostick-v1': dict(
        items={'air', 'crafting_table', 'plank', 'pogo_stick', 'rubber', 'stick', 'tree_log', 'tree_tap', 'wall'},
        goal='pogo_stick',
        items_quantity={'crafting_table': 1, 'tree_log': 5},
        manipulation=['Forward', 'Left', 'Right', 'Break', 'Place_tree_tap', 'Extract_rubber'],
        recipes={'pogo_stick': {'input': {'stick': 4, 'plank': 2, 'rubber': 1}, 'output': {'pogo_stick': 1}},
                 'stick': {'input': {'plank': 2}, 'output': {'stick': 4}},
                 'plank': {'input': {'tree_log': 1}, 'output': {'plank': 4}},
                 'tree_tap': {'input': {'plank': 5, 'stick': 1}, 'output': {'tree_tap': 1}}},
        # item -> (cost when inputs missing, cost when not at crafting_table, cost when crafted)
        craft_costs={'tree_tap': (360.0, 720.0, 7200.0), 'pogo_stick': (480.0, 840.0, 8400.0),
                     'plank': (0, 0, 1200.0), 'stick': (0, 0, 2400.0)},
        craft_reward='intermediate',                       # pogostick_v1_env.py:455
        place=dict(item='tree_tap', near='tree_log'),
        extract=dict(src='tree_tap', near='tree_log', out='rubber', qty=1, consume=False, cost_ok=50000),
    ),
    'NovelGridworld-Bow-v1': dict(
        items={'air', 'bow', 'crafting_table', 'plank', 'stick', 'string', 'tree_log', 'wall', 'wool'},
        goal='bow',
        items_quantity={'crafting_table': 1, 'tree_log': 3, 'wool': 2},
        manipulation=['Forward', 'Left', 'Right', 'Break', 'Extract_string'],
        recipes={'bow': {'input': {'stick': 3, 'string': 3}, 'output': {'bow': 1}},
                 'stick': {'input': {'plank': 2}, 'output': {'stick': 4}},
                 'plank': {'input': {'tree_log': 1}, 'output': {'plank': 4}}},
        craft_costs={'bow': (480.0, 840.0, 8400.0), 'plank': (0, 0, 1200.0), 'stick': (0, 0, 2400.0)},
        craft_reward='done',                               # bow_v1_env.py:424
        place=None,
        extract=dict(src='wool', near=None, out='string', qty=4, consume=True, cost_ok=5000),
    ),
}
# v0 variants: same skeleton, different start items / break rewards / craft reward, and
# Pogostick-v0 pre-places a tree_tap next to a tree_log at reset (pogostick_v0_env.py:44,156-178,312,479; bow_v0_env.py:44,286,424)
_ENV_DEFS['NovelGridworld-Pogostick-v0'] = dict(
    copy.deepcopy(_ENV_DEFS['NovelGridworld-Pogostick-v1']),
    items_quantity={'crafting_table': 1, 'stick': 4, 'plank': 2, 'tree_log': 2}, craft_reward='done',
    break_reward_items=['stick', 'plank'], tap_pass=dict(item='tree_tap', near='tree_log'))
_ENV_DEFS['NovelGridworld-Bow-v0'] = dict(
    copy.deepcopy(_ENV_DEFS['NovelGridworld-Bow-v1']),
    items_quantity={'crafting_table': 1, 'stick': 3, 'string': 3}, craft_reward='intermediate',
    break_reward_items=['stick', 'string'])
ENV_IDS = tuple(_ENV_DEFS)


def set_items_id(items):
    """air = 0, the rest alphabetically (pogostick_v1_env.py:200-212)."""
    items_id = {}
    if 'air' in items:
        items_id['air'] = 0
    for item in sorted(items):
        if item != 'air':
            items_id[item] = len(items_id) if 'air' in items else len(items_id) + 1
    return items_id


class EnvSpec:
    """Mutable host-side description of one environment configuration; `compile()` flattens it.

    Attribute names follow the reference env (items, items_id, items_quantity, entities, actions_id,
    recipes, ...) so novelty injection reads like the reference's table edits."""

    def __init__(self, env_id, map_size=10):
        if env_id not in _ENV_DEFS:
            raise KeyError("unknown env id %r (supported: %s)" % (env_id, ', '.join(ENV_IDS)))
        d = copy.deepcopy(_ENV_DEFS[env_id])
        self.env_id = env_id
        self.class_name = {'NovelGridworld-Pogostick-v1': 'PogostickV1Env', 'NovelGridworld-Bow-v1': 'BowV1Env',
                           'NovelGridworld-Pogostick-v0': 'PogostickV0Env', 'NovelGridworld-Bow-v0': 'BowV0Env'}[env_id]
        self.map_size = int(map_size)
        self.items = set(d['items'])
        self.items_id = set_items_id(self.items)
        self.unbreakable_items = {'air', 'wall'}
        self.goal_item_to_craft = d['goal']
        self.items_quantity = dict(d['items_quantity'])
        self.entities = set()
        self.recipes = d['recipes']
        self.craft_costs = d['craft_costs']
        self.reward_intermediate, self.reward_done = 10, 50
        self.craft_reward = self.reward_intermediate if d['craft_reward'] == 'intermediate' else self.reward_done
        self.place, self.extract = d['place'], d['extract']
        self.break_reward_items = list(d.get('break_reward_items', ['tree_log']))   # Break gives +10 for these (:288 / v0 :312)
        self.tap_pass = d.get('tap_pass')
        self.actions_id = {}
        self.manipulation_actions_id = {a: i for i, a in enumerate(d['manipulation'])}
        self.actions_id.update(self.manipulation_actions_id)
        self.craft_actions_id = {'Craft_' + item: len(self.actions_id) + i
                                 for i, item in enumerate(sorted(self.recipes.keys()))}
        self.actions_id.update(self.craft_actions_id)
        self.select_actions_id = {'Select_' + item: len(self.actions_id) + i
                                  for i, item in enumerate(sorted(self.items ^ self.unbreakable_items))}
        self.actions_id.update(self.select_actions_id)
        self.action_space_n = len(self.actions_id)      # NOT grown by AxeMedium/AddItem
        self.max_items = 20
        # novelty state
        self.axe = None            # dict(item=name, cost=float, qty=int[, required=True]) -> Break override
        self.break_increase = None # BreakIncrease: '' = every block gives 2, or the one item that does
        self.start_inventory = {}  # AxeEasy: item present in the inventory after every reset
        self.additem = None        # dict(item=name, pct=(lo, hi))
        self.replace = None        # ReplaceItem / FireWall: dict(src=name, dst=name, pct=(lo, hi))
        self.fence = None          # Fence / FenceRestriction: dict(item=name, pct=(lo, hi), mode=0|1|2)
        self.fence_pred = None     # FenceRestriction's Break predicate: dict(item=name, mode=0|1|2)
        self.fire_wall = None      # FireWall: item name whose 4-neighbourhood kills the agent
        self.crate = None          # Crate: dict(item='crate', ingredients=[names drawn at injection])
        self.recipe_rewards = {}   # recipe -> reward of a successful craft when it differs from craft_reward (craftable axe)
        self.reset_passes = []     # the shuffled-subset reset passes in the order their wrappers were stacked (innermost first):
                                   # dict(kind='additem'|'replace'|'fence', item=name, src=name or None, pct=(lo, hi))
        self.novelties = []

    # -- table edits used by inject_novelty ---------------------------------------------------
    def add_new_item(self, name):
        """items.add + items_id.setdefault(name, len(items_id)) (pogostick_v1_env.py:497-499)."""
        self.items.add(name)
        self.items_id.setdefault(name, len(self.items_id))

    def add_select_action(self, name):
        """select_actions_id.update({'Select_'+name: len(actions_id)}) (novelty_wrappers.py:131-132)."""
        self.select_actions_id.update({'Select_' + name: len(self.actions_id)})
        self.actions_id.update(self.select_actions_id)

    # -- derived ----------------------------------------------------------------------------------
    @property
    def item_names(self):
        names = [None] * len(self.items_id)
        for k, v in self.items_id.items():
            names[v] = k
        return names

    @property
    def action_names(self):
        names = [None] * len(self.actions_id)
        for k, v in self.actions_id.items():
            names[v] = k
        return names

    @property
    def recipe_names(self):
        return sorted(self.recipes.keys())

    def validate(self):
        S, K, A, R = self.map_size, len(self.items_id), len(self.actions_id), len(self.recipes)
        assert not self.max_items < len(self.items), \
            "Cannot have more than " + str(self.max_items) + " items"          # pogostick_v1_env.py:220
        if not (5 <= S <= MAX_MAP_SIZE):
            raise ValueError("map_size must be in [5, %d], got %d" % (MAX_MAP_SIZE, S))
        if K > MAX_ITEMS or A > MAX_ACTIONS or R > MAX_RECIPES or len(self.items_quantity) > MAX_START:
            raise ValueError("spec exceeds table capacity (K=%d A=%d R=%d)" % (K, A, R))
        for q in self.items_quantity.values():
            if not 0 < q < 256:
                raise ValueError("items_quantity values must be in 1..255")

    def compile(self):
        """Flatten to the `ngw_spec` LUT set."""
        self.validate()
        ids = self.items_id
        s = NgwSpec()
        s.abi_version = ABI_VERSION
        s.map_size, s.n_items, s.n_actions, s.n_recipes = self.map_size, len(ids), len(self.actions_id), len(self.recipes)
        s.reward_step, s.reward_done = -1, self.reward_done
        rnames = self.recipe_names
        for name, a in self.actions_id.items():
            if name in ('Forward', 'Left', 'Right', 'Break', 'Chop', 'Jump'):
                kind, arg = {'Forward': ACT_FORWARD, 'Left': ACT_LEFT, 'Right': ACT_RIGHT, 'Break': ACT_BREAK,
                             'Chop': ACT_CHOP, 'Jump': ACT_JUMP}[name], 0
            elif name.startswith('Place_'):
                kind, arg = ACT_PLACE, ids[name[6:]]
            elif name.startswith('Extract_'):
                kind, arg = ACT_EXTRACT, ids[name[8:]]
            elif name.startswith('Craft_'):
                kind, arg = ACT_CRAFT, rnames.index(name[6:])
            elif name.startswith('Select_'):
                kind, arg = ACT_SELECT, ids[name[7:]]
            else:
                raise ValueError("action %r has no kernel kind" % name)
            s.act_kind[a], s.act_arg[a] = kind, arg
        for name, i in ids.items():
            s.breakable[i] = int(name not in self.unbreakable_items)
            s.entity[i] = int(name in self.entities)
            if self.break_increase is None:
                s.break_reward[i] = self.reward_intermediate if name in self.break_reward_items else -1   # pogostick_v1_env.py:288-289
                s.break_qty[i] = 1
            else:                                         # BreakIncrease.step, novelty_wrappers.py:1444-1456
                s.break_reward[i] = self.reward_intermediate if name not in self.unbreakable_items else -1
                s.break_qty[i] = 2 if self.break_increase in ('', name) else 1
        s.wall_item, s.table_item, s.goal_item = ids['wall'], ids['crafting_table'], ids[self.goal_item_to_craft]
        s.n_entities = len(self.entities)
        for r, name in enumerate(rnames):
            rec = self.recipes[name]
            if len(rec['input']) > MAX_RECIPE_INPUTS:
                raise ValueError("recipe %r has too many inputs" % name)
            for j, (item, q) in enumerate(rec['input'].items()):
                s.recipe_in[r][ids[item]] = q
                s.recipe_in_item[r][j] = ids[item]
            s.recipe_n_in[r] = len(rec['input'])
            s.recipe_out_item[r] = ids[name]
            s.recipe_out_qty[r] = rec['output'][name]
            s.recipe_needs_table[r] = int(len(rec['input']) > 1)
            cm, cn, ck = self.craft_costs.get(name, (0, 0, 0))
            s.cost_missing[r], s.cost_no_table[r], s.cost_ok[r] = cost_code(cm), cost_code(cn), cost_code(ck)
            s.recipe_reward[r] = self.recipe_rewards.get(name, self.craft_reward)
        s.cost_forward, s.cost_turn, s.cost_break = cost_code(27.906975), cost_code(24.0), cost_code(3600.0)
        s.cost_place, s.cost_extract, s.cost_select = cost_code(300.0), cost_code(120.0), cost_code(120.0)
        s.cost_chop, s.cost_jump = cost_code(3600.0 * 1.2), cost_code(27.906975 * 2)
        s.chop_reward = self.reward_intermediate
        if self.place:
            s.place_item, s.place_near = ids[self.place['item']], ids[self.place['near']]
            s.place_reward = self.reward_intermediate
        e = self.extract
        if e:
            s.ext_src, s.ext_near = ids[e['src']], (ids[e['near']] if e['near'] else 0)
            s.ext_out, s.ext_qty, s.ext_consume = ids[e['out']], e['qty'], int(e['consume'])
            s.ext_cost_ok, s.ext_reward = cost_code(e['cost_ok']), self.reward_intermediate
        if self.axe:
            s.axe_item, s.axe_cost, s.axe_qty = ids[self.axe['item']], cost_code(self.axe['cost']), self.axe['qty']
            s.axe_reward = self.reward_intermediate
            s.axe_required = int(bool(self.axe.get('required')))
        s.n_start = len(self.items_quantity)
        for j, (item, q) in enumerate(self.items_quantity.items()):
            s.start_item[j], s.start_qty[j] = ids[item], q
        if len(self.start_inventory) > MAX_INV_START:
            raise ValueError("start_inventory holds more than %d items" % MAX_INV_START)
        s.n_inv_start = len(self.start_inventory)
        for j, (item, q) in enumerate(self.start_inventory.items()):
            s.inv_start_item[j], s.inv_start_qty[j] = ids[item], q
        if self.tap_pass:
            s.tap_item, s.tap_near = ids[self.tap_pass['item']], ids[self.tap_pass['near']]
        if len(self.reset_passes) > MAX_PASSES:
            raise NotImplementedError("more than %d shuffled-subset reset passes in one stack" % MAX_PASSES)
        s.n_passes = len(self.reset_passes)
        for j, ps in enumerate(self.reset_passes):
            s.pass_kind[j] = {'additem': 1, 'replace': 2, 'fence': 3}[ps['kind']]
            s.pass_item[j] = ids[ps['item']]
            s.pass_from[j] = ids[ps['src']] if ps.get('src') else 0
            s.pass_pct_lo[j], s.pass_pct_hi[j] = ps['pct']
        # wrapper nesting of a stack (injection order = inner to outer), as far as the step can tell
        names = [nv[0] for nv in self.novelties]
        last = lambda pred: max([i for i, nv in enumerate(self.novelties) if pred(nv)], default=-1)
        B = last(lambda nv: nv[0] in ('axe', 'axetobreak', 'breakincrease'))           # who handles Break (alone)
        f = last(lambda nv: nv[0] == 'fencerestriction' and nv[1] != 'easy')
        c, w = last(lambda nv: nv[0] == 'crate'), last(lambda nv: nv[0] == 'firewall')
        h = last(lambda nv: nv[0] in ('axe', 'axetobreak') and nv[1] == 'hard')
        fence_on, crate_on = f > B, c > B                      # below the Break handler they are never consulted
        if w >= 0 and w < B:
            s.ext_flags |= 1                                   # NGW_XF_FIRE_SKIP_BREAK
        if w >= 0 and h > w:
            s.fire_skip_recipe = 1 + rnames.index(self.novelties[h][2] + '_axe')
        if crate_on and fence_on and c < f:
            s.ext_flags |= 2                                   # NGW_XF_CRATE_IN_FENCE
        pred = self.fence_pred or self.fence                   # the FenceRestriction predicate looks for ITS fence item
        if pred:
            s.fence_item, s.fence_mode = ids[pred['item']], (pred['mode'] if fence_on or f < 0 else 0)
        if self.fire_wall:
            s.fire_item, s.fire_reward = ids[self.fire_wall], -self.reward_done // 2        # novelty_wrappers.py:1187
        if self.crate and crate_on:
            s.crate_item = ids[self.crate['item']]
            for name in self.crate['ingredients']:
                s.crate_add[ids[name]] += 1
        return s

    # -- host-side decoding of kernel outputs --------------------------------------------------
    def format_message(self, action, code, arg):
        """(code, arg) -> the reference's info['message'] string."""
        names = self.item_names
        if code == MSG_NONE:
            return ''
        if code == MSG_BLOCK_IN_PATH:
            return 'Block in path'                                          # pogostick_v1_env.py:255
        if code == MSG_CANNOT_BREAK:
            return "Cannot break " + names[arg]                             # :292
        if code == MSG_PLACED:
            return "Block " + names[arg] + " placed"                        # :301
        if code == MSG_ALREADY_EXISTS:
            return "Block " + names[arg] + " already exists when trying to place block"   # :309
        if code == MSG_NOT_IN_INVENTORY:
            return "Item not found in inventory"                            # :312, :347
        if code == MSG_EXTRACT_NO_SRC:
            return "No " + self.extract['src'] + " found"                   # :331, bow_v1_env.py:304
        if code == MSG_EXTRACT_NOT_NEAR:
            return "No " + self.extract['near'] + " near " + self.extract['src']   # :328
        if code == MSG_MISSING_ITEMS:
            rec = self.recipes[self.recipe_names[arg >> 8]]['input']        # :432-440
            parts = [str(q) + ' ' + item for j, (item, q) in enumerate(rec.items()) if (arg >> j) & 1]
            return "Missing items: " + ', '.join(parts)
        if code == MSG_NEED_TABLE:
            return 'Need to be in front of crafting_table'                  # :452
        if code == MSG_CRAFTED:
            return 'Crafted ' + names[arg]                                  # :472
        if code == MSG_CANNOT_CHOP:
            return "Cannot chop " + names[arg]                              # novelty_wrappers.py:1308
        if code == MSG_FENCE_RESTRICTION:
            return "Cannot break due to fence restriction"                  # novelty_wrappers.py:944
        if code == MSG_FIRE_WALL:
            return 'You died due to fire_wall'                              # novelty_wrappers.py:1189
        if code == MSG_NEED_AXE:
            return "Cannot break without " + names[arg] + " selected"         # novelty_wrappers.py:591
        raise ValueError("unknown message code %d" % code)

    @staticmethod
    def step_cost(code):
        return STEP_COSTS[code]


def make_spec(env_id, map_size=None):
    return EnvSpec(env_id, 10 if map_size is None else map_size)
