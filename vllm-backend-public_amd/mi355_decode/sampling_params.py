"""Per-request sampling parameters, named as the OpenAI /v1/completions fields
the harness's evaluation jobs send (benchmark-deployer/docs/deployment/post_run.md:46)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class SamplingParams:
    max_tokens: int = 16
    min_tokens: int = 0
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1
    seed: Optional[int] = None
    stop: List[str] = field(default_factory=list)
    stop_token_ids: List[int] = field(default_factory=list)
    ignore_eos: bool = False
    logprobs: Optional[int] = None
    # also score the prompt: log-probability of every prompt token given the ones before it
    # (and the `prompt_logprobs` most likely alternatives); what `echo` + `logprobs` asks for
    prompt_logprobs: Optional[int] = None
    n: int = 1

    def __post_init__(self):
        if self.max_tokens is None or self.max_tokens < 1:
            raise ValueError("max_tokens must be >= 1")
        if self.temperature < 0:
            raise ValueError("temperature must be >= 0")
        if not 0 < self.top_p <= 1:
            raise ValueError("top_p must be in (0, 1]")
        if self.top_k == 0 or self.top_k < -1:
            raise ValueError("top_k must be -1 (off) or >= 1")
        if isinstance(self.stop, str):
            self.stop = [self.stop]
        self.stop = list(self.stop or [])
        if self.logprobs is not None and not 0 <= int(self.logprobs) <= 20:
            raise ValueError("logprobs must be between 0 and 20")
        if self.prompt_logprobs is not None and not 0 <= int(self.prompt_logprobs) <= 20:
            raise ValueError("prompt_logprobs must be between 0 and 20")
        if self.n != 1:
            raise ValueError("n > 1 is not supported")

    @property
    def greedy(self) -> bool:
        return self.temperature == 0.0
