"""MI355X-native per-frame video complexity / quality engine (hot path of
zaki699/Real-Time-Video-Quality-Analysis).  Import as ``rtvqa_amd`` — this
directory's name is not a valid Python identifier, so the top-level
``rtvqa_amd`` package points its ``__path__`` here."""
