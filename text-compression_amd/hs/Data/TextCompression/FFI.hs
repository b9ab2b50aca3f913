{-# LANGUAGE ForeignFunctionInterface #-}
-- | Raw bindings of libtextcomp.so (include/textcomp.h).  NOT COMPILED IN THIS REPOSITORY'S
-- PIPELINE: the build image has no GHC; the same C ABI is exercised by the C++ and Python
-- mirrors and their parity tests.  Every import is `safe`: device calls run for milliseconds.
module Data.TextCompression.FFI where

import Data.Int (Int16, Int32, Int64)
import Data.Word (Word16, Word32, Word64, Word8)
import Foreign.C.String (CString)
import Foreign.Ptr (Ptr)

data TcCtx
data TcFm
data TcComm

foreign import ccall safe "tc_ctx_create"  c_tc_ctx_create  :: Int32 -> Ptr (Ptr TcCtx) -> IO Int32
foreign import ccall safe "tc_ctx_destroy" c_tc_ctx_destroy :: Ptr TcCtx -> IO ()
foreign import ccall safe "tc_last_error"  c_tc_last_error  :: Ptr TcCtx -> IO CString

-- Data.BWT
foreign import ccall safe "tc_bwt_encode"
  c_tc_bwt_encode :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_bwt_decode_sym"
  c_tc_bwt_decode_sym :: Ptr TcCtx -> Ptr Int16 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_suffix_array"
  c_tc_suffix_array :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word32 -> IO Int32
-- Data.MTF
foreign import ccall safe "tc_mtf_encode_sym"
  c_tc_mtf_encode_sym :: Ptr TcCtx -> Ptr Int16 -> Word64 -> Ptr Word16 -> Ptr Int16 -> Ptr Word32 -> IO Int32
foreign import ccall safe "tc_mtf_decode"
  c_tc_mtf_decode :: Ptr TcCtx -> Ptr Word16 -> Word64 -> Ptr Int16 -> Word32 -> Ptr Int16 -> IO Int32
-- Data.RLE
foreign import ccall safe "tc_rle_encode_sym"
  c_tc_rle_encode_sym :: Ptr TcCtx -> Ptr Int16 -> Word64 -> Ptr Word32 -> Ptr Int16 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_rle_decode"
  c_tc_rle_decode :: Ptr TcCtx -> Ptr Word32 -> Ptr Int16 -> Word64 -> Ptr Int16 -> Ptr Word64 -> IO Int32
-- Data.FMIndex
foreign import ccall safe "tc_fm_build"
  c_tc_fm_build :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr (Ptr TcFm) -> IO Int32
foreign import ccall safe "tc_fm_count"
  c_tc_fm_count :: Ptr TcCtx -> Ptr TcFm -> Ptr Word8 -> Ptr Word64 -> Word64 -> Ptr Int64 -> IO Int32
foreign import ccall safe "tc_fm_free" c_tc_fm_free :: Ptr TcFm -> IO ()
foreign import ccall safe "tc_fm_locate"
  c_tc_fm_locate :: Ptr TcCtx -> Ptr TcFm -> Ptr Word8 -> Ptr Word64 -> Word64 -> Ptr Word64 -> Ptr Word64 -> Ptr Word64 -> IO Int32
foreign import ccall unsafe "tc_fm_info"
  c_tc_fm_info :: Ptr TcFm -> Ptr Word64 -> Ptr Word32 -> Ptr Int16 -> Ptr Word64 -> Ptr Word64 -> IO Int32
-- stored / shipped form (no counterpart in the reference): one record, or any length cut into records
foreign import ccall unsafe "tc_container_bound"
  c_tc_container_bound :: Word64 -> Word32 -> Word64
foreign import ccall safe "tc_encode_container"
  c_tc_encode_container :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_encode_container_dev"
  c_tc_encode_container_dev :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_container_info"
  c_tc_container_info :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word64 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_decode_container"
  c_tc_decode_container :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall unsafe "tc_stream_bound"
  c_tc_stream_bound :: Word64 -> Word64 -> Word64
foreign import ccall safe "tc_encode_stream"
  c_tc_encode_stream :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_stream_info"
  c_tc_stream_info :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word64 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_decode_stream"
  c_tc_decode_stream :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr Word8 -> Ptr Word64 -> IO Int32
-- the exchange of the multi-GPU path (one process per GPU; include/textcomp.h, tc_comm_*)
foreign import ccall safe "tc_comm_unique_id"
  c_tc_comm_unique_id :: Ptr TcCtx -> Ptr Word8 -> IO Int32
foreign import ccall safe "tc_comm_create"
  c_tc_comm_create :: Ptr TcCtx -> Ptr Word8 -> Int32 -> Int32 -> Ptr (Ptr TcComm) -> IO Int32
foreign import ccall safe "tc_comm_destroy" c_tc_comm_destroy :: Ptr TcComm -> IO ()
foreign import ccall safe "tc_comm_gather"
  c_tc_comm_gather :: Ptr TcComm -> Int32 -> Ptr Word8 -> Word64 -> Ptr Word8 -> Word64 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_comm_wait" c_tc_comm_wait :: Ptr TcComm -> IO Int32
foreign import ccall safe "tc_comm_broadcast"
  c_tc_comm_broadcast :: Ptr TcComm -> Int32 -> Ptr Word8 -> Word64 -> IO Int32
foreign import ccall unsafe "tc_fm_export_bound"
  c_tc_fm_export_bound :: Ptr TcFm -> Int32 -> Word64
foreign import ccall safe "tc_fm_export_dev"
  c_tc_fm_export_dev :: Ptr TcCtx -> Ptr TcFm -> Int32 -> Ptr Word8 -> Ptr Word64 -> IO Int32
foreign import ccall safe "tc_fm_import_dev"
  c_tc_fm_import_dev :: Ptr TcCtx -> Ptr Word8 -> Word64 -> Ptr (Ptr TcFm) -> IO Int32
