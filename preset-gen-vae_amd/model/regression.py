"""Preset-parameter regression from the latent vector (surface of the reference's ``model/regression.py``).

SURVEY.md §8 row a14 keeps this network only as a *parity output* ("preset-regression MSE"): its three 1024-wide GEMMs
are <1 % of the step and stock rocBLAS GEMMs through torch are explicitly acceptable there, so ``MLPRegression`` is
built from stock ``torch.nn`` modules with the reference's sub-module names (``reg_model.fc1``, ``bn1``, ``drp1``,
``act1`` ... ``fcN``, ``act``; reference regression.py:61-102).  ``FlowRegression`` (nflows) is out of scope.
"""
import torch.nn as nn


class PresetActivation(nn.Module):
    """Per-parameter output activations (reference regression.py:20-53): Hardtanh(0,1) on every output neuron, or - with
    ``cat_softmax_activation`` - Hardtanh on the numerical columns and a softmax over every categorical (one-hot)
    sub-vector.  Stock torch ops; the reference writes into its input in place, this module does not."""

    def __init__(self, idx_helper, numerical_activation=None, cat_softmax_activation=False):
        super().__init__()
        self.idx_helper = idx_helper
        self.numerical_act = nn.Hardtanh(min_val=0.0, max_val=1.0) if numerical_activation is None \
            else numerical_activation
        self.cat_softmax_activation = cat_softmax_activation
        if self.cat_softmax_activation:
            self.categorical_act = nn.Softmax(dim=-1)
            self.num_indexes = list(self.idx_helper.get_numerical_learnable_indexes())
            self.cat_indexes = [list(g) for g in self.idx_helper.get_categorical_learnable_indexes()]

    def forward(self, x):
        if not self.cat_softmax_activation:
            return self.numerical_act(x)
        out = x.clone()
        out[:, self.num_indexes] = self.numerical_act(x[:, self.num_indexes])
        for cat_learnable_indexes in self.cat_indexes:
            out[:, cat_learnable_indexes] = self.categorical_act(x[:, cat_learnable_indexes])
        return out


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
