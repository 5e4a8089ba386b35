"""Novelty injection as spec edits (host logic; no device code here).

Mirrors `inject_novelty(env, novelty_name, difficulty='hard', novelty_arg1='', novelty_arg2='')`
(reference: gym_novel_gridworlds/novelty_wrappers.py:1586-1674): the same argument validation with the
same AssertionError messages, then the same table edits the wrapper constructors perform -
AxeMedium.__init__ :125-134, AxeEasy.__init__ :16-27, AddItem.__init__ :996-1011 - applied to an
`EnvSpec`, which is then recompiled into the kernel LUTs.

In scope: 'axe' (easy / medium) and 'additem'.  The other eleven novelties are listed in
SURVEY.md §8(f) as later rows; they validate like the reference and then raise NotImplementedError.
"""

NOVELTY_NAMES = ['addchop', 'additem', 'addjump', 'axe', 'axetobreak', 'breakincrease', 'crate', 'extractincdec',
                 'fence', 'fencerestriction', 'firewall', 'remapaction', 'replaceitem']
_NEEDS_DIFFICULTY = ['additem', 'axe', 'axetobreak', 'crate', 'fence', 'fencerestriction', 'firewall', 'remapaction',
                     'replaceitem']
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
            raise NotImplementedError("axe/hard (craftable axe, novelty_wrappers.py:216) is outside this build's "
                                      "hot-path scope (SURVEY.md §8(f) row 2)")
        _axe(spec, difficulty, novelty_arg1, breakincrease)
    else:
        raise NotImplementedError("novelty %r is outside this build's hot-path scope (SURVEY.md §8(f))"
                                  % novelty_name)
    spec.novelties.append((novelty_name, difficulty, novelty_arg1, novelty_arg2))
    return spec


def _axe(spec, difficulty, axe_material, breakincrease):
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
    spec.axe = dict(item=axe_name, cost=3600.0 * (0.5 if axe_material == 'wooden' else 0.25),
                    qty=2 if breakincrease == 'true' else 1)


def _add_item(spec, difficulty, item_to_add):
    spec.add_new_item(item_to_add)                                 # :1000-1001 (breakable, not an entity)
    spec.add_select_action(item_to_add)                            # :1003-1004
    spec.additem = dict(item=item_to_add, pct=ADDITEM_PERCENT_RANGE[difficulty])
