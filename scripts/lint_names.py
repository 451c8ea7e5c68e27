"""Poor man's pyflakes (none is installed in the image): reports names that are read in a function but bound nowhere —
not in the function, an enclosing function, the module, or builtins.  GPU sessions are expensive; a NameError found on
the box costs one.      python scripts/lint_names.py [files...]   (default: the package, tests, scripts, bench.py)"""
import ast, builtins, glob, os, sys


SCOPES = (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda, ast.ClassDef, ast.ListComp, ast.SetComp, ast.DictComp,
          ast.GeneratorExp)


def own_nodes(scope):
    """nodes of this scope, not descending into nested scopes (their headers — names, decorators, defaults — are ours)"""
    if isinstance(scope, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
        stack = list(scope.body)
    elif isinstance(scope, ast.Lambda):
        stack = [scope.body]
    else:
        stack = list(ast.iter_child_nodes(scope))
    while stack:
        n = stack.pop()
        yield n
        if isinstance(n, SCOPES):
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                stack.extend(n.args.defaults + [d for d in n.args.kw_defaults if d is not None])
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                stack.extend(n.decorator_list)
            if isinstance(n, ast.ClassDef):
                stack.extend(n.bases)
            continue
        stack.extend(ast.iter_child_nodes(n))


def bindings(scope):
    out = set()
    if isinstance(scope, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
        a = scope.args
        for x in a.args + a.kwonlyargs + a.posonlyargs:
            out.add(x.arg)
        if a.vararg:
            out.add(a.vararg.arg)
        if a.kwarg:
            out.add(a.kwarg.arg)
    for n in own_nodes(scope):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    bad = []

    def visit(scope, chain):
        mine = bindings(scope)
        # comprehension targets are Store names inside the comprehension scope: already collected above
        visible = chain + [mine] if not isinstance(scope, ast.ClassDef) else chain + [mine]
        for n in own_nodes(scope):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load):
                if not any(n.id in sc for sc in visible):
                    bad.append("%s:%d: undefined name %s" % (path, n.lineno, n.id))
            if isinstance(n, SCOPES):
                # a class body is not visible from the functions nested in it
                visit(n, chain if isinstance(scope, ast.ClassDef) else visible)
    visit(tree, [set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__class__"}])
    return bad


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sys.argv[1:] or [f for pat in ("xitorch_amd/**/*.py", "tests/**/*.py", "oracle/*.py", "bench.py",
                                           "__graft_entry__.py", "scripts/*.py")
                             for f in glob.glob(os.path.join(root, pat), recursive=True)]
    bad = [b for f in files for b in check(f)]
    print("\n".join(bad) if bad else "names ok (%d files)" % len(files))
    sys.exit(1 if bad else 0)
