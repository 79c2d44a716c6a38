"""Hygiene of the test suite itself (CPU).  Python keeps the LAST definition of a name: a test pasted twice shadows its
first copy silently (round 5 shipped 307 dead lines that way).  Every tests/*.py is parsed and a
top-level name bound by more than one def / class fails."""
import ast
import glob
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _duplicates(path):
    tree = ast.parse(open(path).read(), path)
    seen, dup = {}, []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            if node.name in seen:
                dup.append('%s: %s defined at lines %d and %d' % (os.path.basename(path), node.name, seen[node.name], node.lineno))
            seen[node.name] = node.lineno
        if isinstance(node, ast.ClassDef):          # methods of a test class shadow each other the same way
            inner = {}
            for sub in node.body:
                if isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    if sub.name in inner:
                        dup.append('%s: %s.%s defined at lines %d and %d' % (os.path.basename(path), node.name, sub.name,
                                                                           inner[sub.name], sub.lineno))
                    inner[sub.name] = sub.lineno
    return dup


def test_no_test_module_defines_a_top_level_name_twice():
    files = sorted(glob.glob(os.path.join(HERE, '*.py')) + glob.glob(os.path.join(HERE, 'golden', 'gen', '*.py')))
    assert len(files) > 20
    dup = [d for f in files for d in _duplicates(f)]
    assert not dup, 'shadowed definitions (only the later one runs):\n' + '\n'.join(dup)


def test_the_duplicate_detector_detects(tmp_path):
    p = tmp_path / 'test_x.py'
    p.write_text('def test_a():\n    pass\n\n\ndef helper():\n    pass\n\n\ndef test_a():\n    assert False\n')
    assert len(_duplicates(str(p))) == 1
