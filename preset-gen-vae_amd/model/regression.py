"""Preset-parameter regression from the latent vector (surface of the reference's ``model/regression.py``).

SURVEY.md §8 row a14 keeps this network only as a *parity output* ("preset-regression MSE"): its three 1024-wide GEMMs
are <1 % of the step and stock rocBLAS GEMMs through torch are explicitly acceptable there, so ``MLPRegression`` is
built from stock ``torch.nn`` modules with the reference's sub-module names (``reg_model.fc1``, ``bn1``, ``drp1``,
``act1`` ... ``fcN``, ``act``; reference regression.py:61-102).  ``FlowRegression`` (nflows) is out of scope.
"""
import torch.nn as nn


class PresetActivation(nn.Module):
    """Hardtanh(0,1) on every output neuron (reference regression.py:20-53 with ``cat_softmax_activation=False``,
    the configuration train.py uses for numeric-only parity; the softmax-on-categorical branch is out of scope)."""

    def __init__(self, idx_helper, numerical_activation=None, cat_softmax_activation=False):
        super().__init__()
        if cat_softmax_activation:
            raise NotImplementedError("softmax activation on categorical sub-vectors is out of scope (SURVEY §8 f4)")
        self.idx_helper = idx_helper
        self.numerical_act = nn.Hardtanh(min_val=0.0, max_val=1.0) if numerical_activation is None \
            else numerical_activation
        self.cat_softmax_activation = False

    def forward(self, x):
        return self.numerical_act(x)


class MLPRegression(nn.Module):
    def __init__(self, architecture, dim_z, idx_helper, dropout_p=0.0, cat_softmax_activation=False):
        super().__init__()
        self.architecture = architecture.split('_')
        self.dim_z = dim_z
        self.idx_helper = idx_helper
        if len(self.architecture) != 1:
            raise NotImplementedError("Arch suffix arguments not implemented yet")
        n_layers, n_neurons = (int(v) for v in self.architecture[0].split('l'))
        self.reg_model = nn.Sequential()
        for l in range(n_layers):
            self.reg_model.add_module('fc{}'.format(l + 1), nn.Linear(dim_z if l == 0 else n_neurons, n_neurons))
            if l < n_layers - 1:  # no BN / dropout in the two last FC layers (regression.py:89-93)
                self.reg_model.add_module('bn{}'.format(l + 1), nn.BatchNorm1d(num_features=n_neurons))
                self.reg_model.add_module('drp{}'.format(l + 1), nn.Dropout(dropout_p))
            self.reg_model.add_module('act{}'.format(l + 1), nn.ReLU())
        self.reg_model.add_module('fc{}'.format(n_layers + 1), nn.Linear(n_neurons, idx_helper.learnable_preset_size))
        self.reg_model.add_module('act', PresetActivation(idx_helper, cat_softmax_activation=cat_softmax_activation))

    def forward(self, z_K):
        return self.reg_model(z_K)
