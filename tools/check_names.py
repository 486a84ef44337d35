"""Flags names a module reads but never binds (a poor man's pyflakes: none is installed here).  Usage: python tools/check_names.py FILES"""
import ast
import builtins
import sys


def check(path):
	tree = ast.parse(open(path).read(), path)
	bound = set(dir(builtins)) | {"__file__", "__name__"}
	for node in ast.walk(tree):
		if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
			bound.add(node.name)
			if not isinstance(node, ast.ClassDef):
				a = node.args
				for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
					bound.add(x.arg)
		elif isinstance(node, ast.Lambda):
			a = node.args
			for x in a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
				bound.add(x.arg)
		elif isinstance(node, (ast.Import, ast.ImportFrom)):
			for al in node.names:
				bound.add((al.asname or al.name).split(".")[0])
		elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
			bound.add(node.id)
		elif isinstance(node, ast.ExceptHandler) and node.name:
			bound.add(node.name)
		elif isinstance(node, (ast.Global, ast.Nonlocal)):
			bound.update(node.names)
	bad = 0
	for node in ast.walk(tree):
		if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound:
			print(f"{path}:{node.lineno}: undefined name {node.id}")
			bad += 1
	return bad


if __name__ == "__main__":
	sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
