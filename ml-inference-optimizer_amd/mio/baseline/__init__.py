from .inference import (BasicInferenceRunner, BlockManager, FusionPattern, FusionRegistry,  # noqa: F401
                        InferenceRunner, PagedKVCache, SequenceMetadata, convert_to_flash_attention,
                        create_inference_runner, fusion_registry)
