"""Signatures (parameter names, order, defaults) of the reference's public callables on the hot path, read from the source
text with `ast` (the modules cannot be imported without TensorFlow) into tests/golden/api.json.  Data only.

    python tests/golden/gen/make_api_golden.py"""
import ast
import json
import os

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'api.json')
TARGETS = [('SynthSR/training.py', 'training'), ('SynthSR/fine_tuning_with_adversary.py', 'training'),
           ('SynthSR/fine_tuning_with_adversary.py', 'make_discriminator'), ('SynthSR/brain_generator.py', 'BrainGenerator.__init__'),
           ('SynthSR/labels_to_image_model.py', 'labels_to_image_model'), ('SynthSR/model_inputs.py', 'build_model_inputs'),
           ('ext/neuron/models.py', 'unet'), ('SynthSR/estimate_priors.py', 'build_intensity_stats'),
           ('SynthSR/estimate_priors.py', 'sample_intensity_stats_from_image'),
           ('SynthSR/estimate_priors.py', 'sample_intensity_stats_from_single_dataset'),
           ('ext/lab2im/utils.py', 'load_volume'), ('ext/lab2im/utils.py', 'save_volume'),
           ('ext/lab2im/utils.py', 'get_volume_info'), ('ext/lab2im/utils.py', 'get_list_labels'),
           ('ext/lab2im/edit_volumes.py', 'align_volume_to_ref'), ('ext/lab2im/edit_volumes.py', 'resample_volume'),
           ('ext/lab2im/edit_volumes.py', 'resample_volume_like'), ('ext/lab2im/edit_volumes.py', 'rescale_volume')]


def find(tree, dotted):
    node = tree
    for part in dotted.split('.'):
        node = next(n for n in ast.walk(node) if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == part)
    return node


def signature(fn):
    args = [a.arg for a in fn.args.args]
    defaults = [None] * (len(args) - len(fn.args.defaults)) + list(fn.args.defaults)
    out = []
    for name, d in zip(args, defaults):
        if name == 'self':
            continue
        if d is None:
            out.append([name, '<required>'])
        else:
            try:
                out.append([name, repr(ast.literal_eval(d))])
            except ValueError:
                out.append([name, '<expr>'])
    return out


def main():
    api = {}
    for path, name in TARGETS:
        tree = ast.parse(open(os.path.join(REF, path)).read())
        api['%s:%s' % (path, name)] = signature(find(tree, name))
    with open(OUT, 'w') as f:
        json.dump(api, f, indent=1)
    print({k: len(v) for k, v in api.items()})


if __name__ == '__main__':
    main()
