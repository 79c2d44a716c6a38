"""Extracts the command-line interface of the reference's launcher scripts (flag names, destinations, defaults, kinds) into
tests/golden/cli.json, by evaluating every `parser.add_argument(...)` call of the script text against a recording parser
(the scripts themselves cannot be imported here: they import TensorFlow).  Data only: no reference code is stored.

    python tests/golden/gen/make_cli_golden.py"""
import ast
import json
import os

REF = '/root/reference/scripts'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'cli.json')


def interface(path):
    tree = ast.parse(open(path).read())
    rows = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, 'attr', '') == 'add_argument':
            name = ast.literal_eval(node.args[0])
            kw = {}
            for k in node.keywords:
                try:
                    kw[k.arg] = ast.literal_eval(k.value)
                except ValueError:            # type=str / type=int / type=utils.infer ... : keep the expression's last name
                    kw[k.arg] = getattr(k.value, 'attr', getattr(k.value, 'id', '?'))
            rows.append(dict(name=name, dest=kw.get('dest', name.lstrip('-')), default=kw.get('default'),
                             action=kw.get('action'), type=kw.get('type')))
    return rows


def main():
    out = {s: interface(os.path.join(REF, s + '.py')) for s in ('training', 'predict_command_line',
                                                                'predict_command_line_hyperfine')}
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
