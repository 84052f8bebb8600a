"""CPU restatement of the reference's Transformer classifier for UrbanSound8K log-mels.

TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Follows /root/reference/.charles/spectrogram.py:944-1038 (`TransformerUrbanSound8KClassifier.__init__` / `.forward`): the
reference composes stock `torch.nn` modules, so this restatement composes the same modules in the same order -- the
arithmetic is torch's own (Linear -> CLS token -> learnable positions -> nn.TransformerEncoder of post-LN GELU layers ->
LayerNorm -> CLS -> Linear / ReLU / Linear).  The reference file itself cannot be imported here (module-level imports of
torchaudio, dotenv, colorlog, wandb, seaborn: SURVEY.md §8c); no reference fixture exists for it, so parity is pinned to
this restatement.
"""
import torch


class ReferenceTransformerClassifier(torch.nn.Module):
    def __init__(self, n_classes=10, n_mels=128, dim=128, depth=2, heads=4, mlp_dim=256, dropout=0.1):
        super().__init__()
        self.dim = dim
        self.input_proj = torch.nn.Linear(n_mels, dim)                       # :965
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, dim))          # :968-969
        torch.nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.pos_embed = None                                                # :972 (created in forward, :1017-1021)
        self.n_frames = None
        layer = torch.nn.TransformerEncoderLayer(d_model=dim, nhead=heads, dim_feedforward=mlp_dim, dropout=dropout,
                                                 activation="gelu", batch_first=True)     # :975-982
        self.encoder = torch.nn.TransformerEncoder(layer, num_layers=depth)  # :983
        self.dropout = torch.nn.Dropout(dropout)
        self.norm = torch.nn.LayerNorm(dim)                                  # :986
        self.head = torch.nn.Sequential(torch.nn.Linear(dim, mlp_dim), torch.nn.ReLU(), torch.nn.Dropout(dropout),
                                        torch.nn.Linear(mlp_dim, n_classes))              # :989-994

    def features(self, x):
        x = x.transpose(1, 2)                                                # :1005
        B, T, _ = x.shape
        x = self.input_proj(x)                                               # :1009
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)          # :1012-1015
        if self.pos_embed is None or self.n_frames != T:                     # :1017-1021
            self.n_frames = T
            self.pos_embed = torch.nn.Parameter(torch.zeros(1, T + 1, self.dim))
            torch.nn.init.trunc_normal_(self.pos_embed, std=0.02)
        x = self.dropout(x + self.pos_embed)                                 # :1023-1024
        return self.norm(self.encoder(x))[:, 0]                              # :1027-1031

    def forward(self, x):
        return self.head(self.features(x))                                   # :1033
