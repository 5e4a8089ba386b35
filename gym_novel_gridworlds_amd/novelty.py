"""Novelty injection as spec edits (host logic; no device code here).

Mirrors `inject_novelty(env, novelty_name, difficulty='hard', novelty_arg1='', novelty_arg2='')`
(reference: gym_novel_gridworlds/novelty_wrappers.py:1586-1674): the same argument validation with the
same AssertionError messages, then the same table edits the wrapper constructors perform -
AxeMedium.__init__ :125-134, AxeEasy.__init__ :16-27, AddItem.__init__ :996-1011 - applied to an
`EnvSpec`, which is then recompiled into the kernel LUTs.

In scope: SURVEY.md §8(a) 'axe' (easy / medium) and 'additem'; §8(f) row 2 (LUT-only novelties) 'breakincrease',
'extractincdec', 'axetobreak', 'remapaction', 'addchop' and 'addjump'; 'axe' / 'axetobreak' hard (craftable axe); §8(f) row 3
(reset-time map edits + step predicates) 'fence', 'fencerestriction', 'replaceitem', 'firewall' and 'crate' - i.e. every
name `inject_novelty` accepts.
"""

NOVELTY_NAMES = ['addchop', 'additem', 'addjump', 'axe', 'axetobreak', 'breakincrease', 'crate', 'extractincdec',
                 'fence', 'fencerestriction', 'firewall', 'remapaction', 'replaceitem']
_NEEDS_DIFFICULTY = ['additem', 'axe', 'axetobreak', 'crate', 'fence', 'fencerestriction', 'firewall', 'remapaction',
                     'replaceitem']
MAX_PASSES = 4                  # = include/ngw.h NGW_MAX_PASSES (spec.MAX_PASSES)
ADDITEM_PERCENT_RANGE = {'easy': (1, 10), 'medium': (10, 20), 'hard': (20, 30)}     # novelty_wrappers.py:1006-1011


def apply_novelty(spec, novelty_name, difficulty='hard', novelty_arg1='', novelty_arg2=''):
    """Edits `spec` (an EnvSpec) in place and returns it."""
    assert novelty_name in NOVELTY_NAMES, "novelty_name must be one of " + str(NOVELTY_NAMES)        # :1590
    if novelty_name in _NEEDS_DIFFICULTY:
        assert difficulty in ['easy', 'medium', 'hard'], "difficulty must be one of 'easy', 'medium', 'hard'"   # :1592

    if novelty_name == 'additem':
        assert novelty_arg1, "For additem novelty, novelty_arg1 (name of the item to add) is needed"    # :1597
        _add_item(spec, difficulty, novelty_arg1)
    elif novelty_name == 'axe':
        assert novelty_arg1 in ['wooden', 'iron'], \
            "For axe novelty, novelty_arg1 (attribute of axe, e.g. wooden, iron) is needed"             # :1603
        breakincrease = 'false'
        if novelty_arg2:
            assert novelty_arg2 in ['true', 'false'], \
                "For axe novelty, novelty_arg2 (breakincrease) must be 'true' or 'false'"               # :1607
            breakincrease = novelty_arg2
        if difficulty == 'hard':
            _axe_hard(spec, novelty_arg1, breakincrease)
        else:
            _axe(spec, difficulty, novelty_arg1, breakincrease)
        spec.break_increase = None        # stacked wrappers: the OUTERMOST Break-overriding wrapper handles Break alone (none of
                                          # AxeEasy/Medium/Hard, AxetoBreak*, BreakIncrease delegates Break to the wrapped env)
    elif novelty_name == 'axetobreak':
        assert novelty_arg1 in ['wooden', 'iron'], \
            "For axe novelty, novelty_arg1 (attribute of axe, e.g. wooden, iron) is needed"             # :1623
        if difficulty == 'hard':
            _axe_hard(spec, novelty_arg1, 'false', required=True)
        else:
            _axe(spec, difficulty, novelty_arg1, 'false', required=True)
        spec.break_increase = None
    elif novelty_name == 'breakincrease':
        if novelty_arg1 and novelty_arg1 not in spec.items:
            # the reference's assert message dereferences env.itemtobreakmore, which does not exist (:1634, SURVEY
            # appendix #11): what surfaces is this AttributeError, not an AssertionError
            raise AttributeError("'%s' object has no attribute 'itemtobreakmore'" % spec.class_name)
        spec.break_increase = novelty_arg1                         # BreakIncrease.__init__ :1421-1424 ('' = every block)
        spec.axe = None                   # an axe wrapper below it no longer sees Break (its item, entity, Select action and
                                          # start inventory stay)
    elif novelty_name == 'extractincdec':
        assert novelty_arg1 in ['increase', 'decrease'], \
            "For extractincdec novelty, novelty_arg1 ('increase', 'decrease') is needed"                # :1642
        assert spec.env_id != 'NovelGridworld-Bow-v0', "There is nothing to extract in NovelGridworld-Bow-v0"
        if spec.env_id == 'NovelGridworld-Bow-v1':
            assert novelty_arg1 == 'decrease', "In NovelGridworld-Bow-v1, increasing string extraction will not benefit " \
                                               "as only 3 string are needed"                          # :1648
        assert not spec.env_id.startswith('NovelGridworld-Pogostick'), "In NovelGridworld-Pogostick, you should not use " \
            "extractincdec novelty because rubber extraction cannot be decreased, and increasing rubber extraction will" \
            " not benefit as only 1 rubber is needed"                                                  # :1651
        # ExtractIncDec.step :1526-1530: 4 * 2 or 4 // 2 strings per wool; everything else as Extract_string
        spec.extract['qty'] = spec.extract['qty'] * 2 if novelty_arg1 == 'increase' else spec.extract['qty'] // 2
    elif novelty_name == 'remapaction':
        _remap_action_difficulty(spec, difficulty)
    elif novelty_name in ('addchop', 'addjump'):
        # AddChopAction.__init__ :1273-1278 / AddJumpAction.__init__ :1345-1350: a new manipulation action at the end
        name = 'Chop' if novelty_name == 'addchop' else 'Jump'
        spec.manipulation_actions_id[name] = len(spec.actions_id)
        spec.actions_id.update(spec.manipulation_actions_id)
        spec.action_space_n = len(spec.actions_id)                 # these wrappers DO grow their action_space (:1278, :1350)
    elif novelty_name == 'crate':
        _crate(spec, difficulty)
    elif novelty_name == 'fence':
        assert novelty_arg1, "For fence novelty, novelty_arg1 (attribute of fence, e.g. oak, jungle) is needed"   # :1660
        _fence(spec, difficulty, novelty_arg1, mode=0)
    elif novelty_name == 'fencerestriction':
        assert novelty_arg1, \
            "For fencerestriction novelty, novelty_arg1 (attribute of fence, e.g. oak, jungle) is needed"       # :1664
        # FenceRestriction.__init__ :901: the fences are ALWAYS laid out by Fence(env, 'medium', ...); the difficulty
        # only selects the Break predicate (:906, :927-941)
        _fence(spec, 'medium', novelty_arg1, mode={'easy': 0, 'medium': 1, 'hard': 2}[difficulty], restriction=True)
    elif novelty_name == 'firewall':
        _replace_item(spec, difficulty, 'wall', 'fire_wall')       # FireWall.__init__ :1159
        spec.fire_wall = 'fire_wall'
    elif novelty_name == 'replaceitem':
        assert novelty_arg1 and novelty_arg2, "For replaceitem novelty, novelty_arg1 (Item to replace) and novelty_arg2" \
                                              "(Item to replace with) are needed"                       # :1672
        _replace_item(spec, difficulty, novelty_arg1, novelty_arg2)
    kind = {'additem': 'additem', 'crate': 'additem', 'replaceitem': 'replace', 'firewall': 'replace', 'fence': 'fence',
            'fencerestriction': 'fence'}.get(novelty_name)
    if kind:        # reset passes run innermost wrapper first = injection order (Wrapper.reset calls env.reset() first); any number of
                    # passes of one kind may be stacked (additem + crate, fence + fencerestriction, replaceitem + firewall)
        if kind == 'fence' and any(ps['kind'] == 'replace' and ps['src'] == 'wall' for ps in spec.reset_passes):
            # the fence pass would pick cells of the (replaced) border ring, and add_fence_around (pogostick_v1_env.py:524-536)
            # then indexes row / column -1 and S: numpy wraps the first around and raises IndexError on the second - not a
            # behaviour to reproduce; refuse the stack instead of editing cells outside the map
            raise IndexError("fence after a wall-replacing novelty (%s): the reference's add_fence_around indexes outside the map"
                             % ', '.join(n[0] for n in spec.novelties if n[0] in ('firewall', 'replaceitem')))
        if len(spec.reset_passes) >= MAX_PASSES:
            raise NotImplementedError("more than %d shuffled-subset reset passes in one stack" % MAX_PASSES)
        src = {'additem': spec.additem, 'replace': spec.replace, 'fence': spec.fence}[kind]
        spec.reset_passes.append(dict(kind=kind, item=src.get('dst', src.get('item')), src=src.get('src'), pct=tuple(src['pct'])))
    spec.novelties.append((novelty_name, difficulty, novelty_arg1, novelty_arg2))
    return spec


def _remap_action(actions_id, start_action_id):
    """Shuffle the action names with the GLOBAL numpy stream until the mapping changes (pogostick_v1_env.py:476-493),
    so `np.random.seed(s); inject_novelty(env, 'remapaction', ...)` yields the reference's permutation."""
    import numpy as np
    while True:
        actions = list(actions_id.keys())
        np.random.shuffle(actions)
        actions_id_new = {actions[i - start_action_id]: i for i in range(start_action_id, start_action_id + len(actions))}
        if actions_id != actions_id_new:
            actions_id = actions_id_new
            print("New remapped actions: ", actions_id)
            break
    return actions_id


def _remap_action_difficulty(spec, difficulty):
    """remap_action_difficulty, novelty_wrappers.py:1203-1227 (the LimitActions branch lives in wrappers.LimitActions)."""
    if difficulty == 'easy':
        spec.manipulation_actions_id = _remap_action(spec.manipulation_actions_id, 0)
        spec.actions_id.update(spec.manipulation_actions_id)
    elif difficulty == 'medium':
        spec.manipulation_actions_id = _remap_action(spec.manipulation_actions_id, 0)
        spec.craft_actions_id = _remap_action(spec.craft_actions_id, len(spec.manipulation_actions_id))
        spec.actions_id.update(spec.manipulation_actions_id)
        spec.actions_id.update(spec.craft_actions_id)
    else:
        remapped = _remap_action(spec.actions_id, 0)
        spec.actions_id.clear()
        spec.actions_id.update(remapped)                           # keep the dict object: the env adapters alias it
        spec.craft_actions_id = {a: spec.actions_id[a] for a in spec.actions_id if a.startswith('Craft')}
        spec.select_actions_id = {a: spec.actions_id[a] for a in spec.actions_id if a.startswith('Select')}


def _axe(spec, difficulty, axe_material, breakincrease, required=False):
    axe_name = axe_material + '_axe'                               # :128
    spec.add_new_item(axe_name)                                    # add_new_items / items_id.setdefault
    if difficulty == 'medium':
        spec.items_quantity.update({axe_name: 1})                  # pogostick_v1_env.py:500 - axe lies on the map
    else:
        spec.start_inventory = {axe_name: 1}                       # AxeEasy.reset :33 - axe starts in the inventory
    spec.entities.add(axe_name)                                    # :130 / :23
    spec.add_select_action(axe_name)                               # :131-132 (action_space is NOT grown)
    # Break override :144-183: selected axe -> cost 3600*0.5 (wooden) / 3600*0.25 (iron), reward +10 for any block,
    # +2 blocks with breakincrease; without it a breakable block gives -1 even for tree_log.
    # AxetoBreak* (required=True, :439-625): the same, but without the selected axe Break fails with
    # 'Cannot break without <axe> selected'.
    spec.axe = dict(item=axe_name, cost=3600.0 * (0.5 if axe_material == 'wooden' else 0.25),
                    qty=2 if breakincrease == 'true' else 1, required=required)


def _axe_hard(spec, axe_material, breakincrease, required=False):
    """AxeHard.__init__ (novelty_wrappers.py:225-258) / AxetoBreakHard.__init__ + reset (:636-672): the axe has to be
    crafted ({'stick': 2, 'plank': 3} wooden / {'stick': 2, 'iron': 3} iron, at the crafting_table).  AxeHard scatters the
    ingredients on the map; AxetoBreakHard puts them in the inventory at every reset.  Craft costs of the axe, from the
    wrappers' own craft() (:288-353 / :694-758): 0 (int) when inputs are missing, 600.0 away from the table, 6000.0."""
    axe_name = axe_material + '_axe'
    spec.add_new_item(axe_name)
    spec.entities.add(axe_name)
    axe_recipe = {'stick': 2, 'plank': 3} if axe_material == 'wooden' else {'stick': 2, 'iron': 3}
    if required:
        for item in axe_recipe:                                    # :651-655 - ingredients start in the inventory
            if item not in spec.items:
                spec.add_new_item(item)
        spec.start_inventory = dict(axe_recipe)                    # reset :667-670
    else:
        for item in axe_recipe:                                    # :241-250 - ingredients are placed on the map
            if item in spec.items:
                spec.items_quantity.update({item: spec.items_quantity.get(item, 0) + axe_recipe[item]})
            else:
                spec.add_new_item(item)                            # add_new_items (pogostick_v1_env.py:495-501)
                spec.items_quantity.update({item: axe_recipe[item]})
    spec.recipes.update({axe_name: {'input': axe_recipe, 'output': {axe_name: 1}}})
    spec.craft_costs[axe_name] = (0, 600.0, 6000.0)
    spec.recipe_rewards[axe_name] = spec.reward_intermediate       # the wrappers' craft() (:331 / :737), also in Bow
    if not required:
        spec.craft_actions_id.update({'Craft_' + axe_name: len(spec.actions_id)})   # :252 (AxetoBreakHard leaves this table alone, :658)
    spec.actions_id.update({'Craft_' + axe_name: len(spec.actions_id)})
    spec.add_select_action(axe_name)
    spec.base_action_space_n = len(spec.actions_id)                # only the BASE env's action_space is re-made (:256 / :662);
                                                                   # the wrapper keeps the copy it took before (action_space_n)
    spec.axe = dict(item=axe_name, cost=3600.0 * (0.5 if axe_material == 'wooden' else 0.25),
                    qty=2 if breakincrease == 'true' else 1, required=required)


FENCE_PERCENT_RANGE = {'easy': (20, 50), 'medium': (50, 90), 'hard': (90, 100)}     # novelty_wrappers.py:861-866
REPLACE_PERCENT_RANGE = {'easy': (5, 20), 'medium': (40, 90), 'hard': (99, 100)}    # :1121-1126
CRATE_PERCENT_RANGE = {'easy': (99, 100), 'medium': (50, 90), 'hard': (10, 50)}     # :1047-1052


def _fence(spec, difficulty, fence_material, mode, restriction=False):
    """Fence.__init__ (novelty_wrappers.py:852-866): new item <material>_fence + its Select action; the fences appear in
    the reset pass (:867-889).  `mode` is FenceRestriction's Break predicate (0 = none)."""
    fence_name = fence_material + '_fence'
    spec.add_new_item(fence_name)
    spec.add_select_action(fence_name)
    spec.fence = dict(item=fence_name, pct=FENCE_PERCENT_RANGE[difficulty], mode=mode)
    if restriction:                                              # the Break predicate belongs to the FenceRestriction wrapper and ITS fence item:
        spec.fence_pred = dict(item=fence_name, mode=mode)        # a plain Fence wrapper stacked on top later does not touch it


def _replace_item(spec, difficulty, item_to_replace, item_to_replace_with):
    """ReplaceItem.__init__ (:1100-1126)."""
    assert item_to_replace in spec.items_id, "Item to replace (" + item_to_replace + ") is not in the original map"
    assert item_to_replace_with not in spec.items_id, "Item to replace with (" + item_to_replace_with + \
                                                      ") should be a new item"
    spec.add_new_item(item_to_replace_with)
    spec.add_select_action(item_to_replace_with)
    if item_to_replace == 'wall':
        spec.unbreakable_items.add(item_to_replace_with)           # :1118-1119
    spec.replace = dict(src=item_to_replace, dst=item_to_replace_with, pct=REPLACE_PERCENT_RANGE[difficulty])


def _crate(spec, difficulty):
    """Crate.__init__ (:1043-1068): AddItem(env, 'easy', 'crate') + the multiset of goal-recipe ingredients a crate holds,
    drawn HERE from the global numpy stream with the reference's calls (randint, then choice until the quota is met), so
    `np.random.seed(s); inject_novelty(env, 'crate', ...)` gives the reference's crate."""
    import numpy as np
    _add_item(spec, 'easy', 'crate')
    lo, hi = CRATE_PERCENT_RANGE[difficulty]
    item_percent = np.random.randint(low=lo, high=hi, size=1)[0]
    goal_in = spec.recipes[spec.goal_item_to_craft]['input']
    total_ingredients = sum(goal_in.values())
    ingredients = list(goal_in)
    crate_ingredients_num = int(np.ceil((item_percent / 100) * total_ingredients))
    crate_ingredients = []
    while crate_ingredients_num:
        item = np.random.choice(ingredients, size=1)[0]
        if crate_ingredients.count(item) < goal_in[item]:
            crate_ingredients.append(str(item))
            crate_ingredients_num -= 1
    spec.crate = dict(item='crate', ingredients=crate_ingredients)


def _add_item(spec, difficulty, item_to_add):
    spec.add_new_item(item_to_add)                                 # :1000-1001 (breakable, not an entity)
    spec.add_select_action(item_to_add)                            # :1003-1004
    spec.additem = dict(item=item_to_add, pct=ADDITEM_PERCENT_RANGE[difficulty])
