from .knn import FlatIPIndex, knn_topk_ip  # noqa: F401
from .reranker import maximal_marginal_relevance, reranker_random, reranker_top_k  # noqa: F401
from .retriever import (RandomRetrievalDatasetWrapper, RetrievalDatasetWrapper, Retriever, coarse_saliency, cross_dataset_table, load_cache_table, merge_retrieval_cache,  # noqa: F401
                        merged_vectors, table_path)
from .embed import coarse_saliency_batch, layout_features, pool_cosine, rerank_tables  # noqa: F401
from .faiss_io import read_flat_index, write_flat_index  # noqa: F401
from .sharded import merge_topk, query_block, search_index_sharded, search_query_sharded  # noqa: F401
