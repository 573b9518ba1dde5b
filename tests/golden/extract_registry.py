"""Pins the level DATA and the id table the reference holds as source constants (build container only).

    python tests/golden/extract_registry.py        # writes tests/golden/mg_registry.json

`envs/multigrid/*.py` cannot be imported here (gym / gym_minigrid / baselines are absent), so the files are read with
`ast` and nothing is executed.  Extracted, per module: every `register(...)` call (id, entry-point class,
max_episode_steps) and, per class, its bases, the literal defaults of `__init__`, the literal keyword arguments of its
`super().__init__(...)` call and the literal local assignments of `__init__` (the maze bit maps, start / goal cells
and the `np.random.choice` lists of the corridor mazes).  Only data is written; tests/test_registry_vs_reference.py
compares the repo's registry and maze tables with it.
"""
import ast
import json
import os

REF = "/root/reference/envs/multigrid"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mg_registry.json")
FILES = ["adversarial.py", "maze.py", "mst_maze.py", "cluttered.py", "crossing.py", "fourrooms.py"]


def literal(node):
    """Python literal of an expression; np.array(x) -> x; np.random.choice(xs) -> {"choice": xs}; a bare name ->
    {"name": id}; arithmetic on names (2*size*size) -> {"expr": source}."""
    if isinstance(node, ast.Call):
        fn = ast.unparse(node.func)
        if fn in ("np.array", "numpy.array") and node.args:
            return literal(node.args[0])
        if fn == "np.random.choice" and node.args:
            return {"choice": literal(node.args[0])}
        return {"expr": ast.unparse(node)}
    if isinstance(node, ast.Name):
        return {"name": node.id}
    if isinstance(node, (ast.List, ast.Tuple)):
        return [literal(e) for e in node.elts]
    try:
        return ast.literal_eval(node)
    except Exception:
        return {"expr": ast.unparse(node)}


def entry_class(node):
    """`module_path + ':Cls'` or 'pkg.mod:Cls' -> 'Cls'."""
    if isinstance(node, ast.BinOp) and isinstance(node.right, ast.Constant):
        return node.right.value.lstrip(":")
    if isinstance(node, ast.Constant):
        return node.value.split(":")[-1]
    return ast.unparse(node)


def extract(path):
    tree = ast.parse(open(path).read())
    classes, regs = {}, []
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            info = {"bases": [ast.unparse(b).split(".")[-1] for b in node.bases], "init_defaults": {}, "super_kwargs": {},
                    "assigns": {}}
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "__init__":
                    args = item.args.args[1:]
                    defaults = item.args.defaults
                    for a, d in zip(args[len(args) - len(defaults):], defaults):
                        info["init_defaults"][a.arg] = literal(d)
                    for sub in ast.walk(item):
                        if isinstance(sub, ast.Assign) and len(sub.targets) == 1 and isinstance(sub.targets[0], ast.Name):
                            info["assigns"].setdefault(sub.targets[0].id, literal(sub.value))   # first assignment wins
                        if (isinstance(sub, ast.Call) and isinstance(sub.func, ast.Attribute) and sub.func.attr == "__init__"
                                and isinstance(sub.func.value, ast.Call) and ast.unparse(sub.func.value.func) == "super"):
                            for kw in sub.keywords:
                                if kw.arg is not None:
                                    info["super_kwargs"][kw.arg] = literal(kw.value)
            classes[node.name] = info
        elif isinstance(node, ast.Expr) and isinstance(node.value, ast.Call):
            call = node.value
            if ast.unparse(call.func) in ("register", "register.register"):
                kw = {k.arg: k.value for k in call.keywords}
                env_id = kw.get("id", kw.get("env_id"))
                regs.append({"id": ast.literal_eval(env_id), "cls": entry_class(kw["entry_point"]),
                             "entry_point": ast.unparse(kw["entry_point"]),
                             "max_episode_steps": ast.literal_eval(kw["max_episode_steps"]) if "max_episode_steps" in kw else None})
    return {"classes": classes, "registrations": regs}


def main():
    out = {name: extract(os.path.join(REF, name)) for name in FILES}
    # MazeEnv's default bit map is assigned to self.bit_map inside an `if`: take it from the source by position
    tree = ast.parse(open(os.path.join(REF, "maze.py")).read())
    for node in ast.walk(tree):
        if (isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) and node.targets[0].attr == "bit_map"
                and isinstance(node.value, ast.Call)):
            out["maze.py"]["classes"]["MazeEnv"]["default_bit_map"] = literal(node.value)
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id.startswith("default_"):
            out["maze.py"]["classes"]["MazeEnv"].setdefault("defaults", {})[node.targets[0].id] = literal(node.value)
    # adversarial.py: editor action spaces (module-level dict)
    tree = ast.parse(open(os.path.join(REF, "adversarial.py")).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id == "EDITOR_ACTION_SPACES":
            out["adversarial.py"]["EDITOR_ACTION_SPACES"] = {k: {str(i): v for i, v in d.items()}
                                                              for k, d in ast.literal_eval(node.value).items()}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n = sum(len(v["registrations"]) for v in out.values())
    print(f"{OUT}: {n} registered ids, {sum(len(v['classes']) for v in out.values())} classes")


if __name__ == "__main__":
    main()
