-- | Drop-in bodies for the Word8 / ByteString helpers of Data.BWT, Data.MTF, Data.RLE and
-- Data.FMIndex: same names, same types, the L2 `Seq (Maybe a)` kernels replaced by calls into
-- libtextcomp.so.  A maintainer re-exports these from the original modules (export lists
-- unchanged).  NOT COMPILED HERE (no GHC in the build image); kept thin and mechanical.
-- `Maybe Word8` <-> Int16, -1 = Nothing.
module Data.TextCompression.GPU
  ( bytestringToBWT, bytestringFromWord8BWT
  , bytestringBWTToMTFB, bytestringBWTFromMTFB
  , bytestringBWTToRLEB, bytestringBWTFromRLEB
  , bytestringFMIndexCountS, bytestringFMIndexCountP
  ) where

import Control.Exception (bracket)
import Control.Monad (when)
import qualified Data.ByteString as BS
import qualified Data.ByteString.Char8 as BSC8
import qualified Data.ByteString.Unsafe as BSU
import Data.BWT.Internal (BWT (..))
import Data.Foldable (toList)
import Data.Int (Int16)
import Data.MTF.Internal (MTF (..))
import Data.RLE.Internal (RLE (..))
import qualified Data.Sequence as DS
import Data.Sequence (Seq)
import Data.Word (Word8)
import Foreign
import Foreign.C.String (peekCString)
import System.IO.Unsafe (unsafePerformIO)

import Data.TextCompression.FFI

withCtx :: (Ptr TcCtx -> IO a) -> IO a
withCtx = bracket open c_tc_ctx_destroy
  where open = alloca $ \pp -> do
          rc <- c_tc_ctx_create 0 pp
          when (rc /= 0) (ioError (userError "textcomp: no usable HIP device"))
          peek pp

-- TC_ERR_MALFORMED (-3) stands for the `error` the reference raises itself
-- (fromJust / DS.index / read); everything else is an infrastructure failure.
check :: Ptr TcCtx -> Int32 -> IO ()
check ctx rc = when (rc /= 0) $ do
  msg <- c_tc_last_error ctx >>= peekCString
  if rc == (-3) then errorWithoutStackTrace msg else ioError (userError msg)

symOf :: Maybe Word8 -> Int16
symOf = maybe (-1) fromIntegral

ofSym :: Int16 -> Maybe Word8
ofSym s = if s < 0 then Nothing else Just (fromIntegral s)

-- | BWT.hs:68-70
bytestringToBWT :: BS.ByteString -> BWT Word8
bytestringToBWT bs
  | BS.null bs = BWT DS.Empty
  | otherwise  = unsafePerformIO $ withCtx $ \ctx ->
      BSU.unsafeUseAsCStringLen bs $ \(p, n) ->
      allocaBytes (n + 1) $ \l -> alloca $ \prim -> do
        c_tc_bwt_encode ctx (castPtr p) (fromIntegral n) l prim >>= check ctx
        pr <- fromIntegral <$> peek prim
        bytes <- peekArray (n + 1) l
        pure . BWT . DS.fromList $ [ if j == pr then Nothing else Just b | (j, b) <- zip [0 :: Int ..] bytes ]

-- | BWT.hs:108-110
bytestringFromWord8BWT :: BWT Word8 -> BS.ByteString
bytestringFromWord8BWT (BWT s)
  | DS.null s = BS.empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map symOf (toList s)) $ \n syms ->
      allocaBytes n $ \out -> alloca $ \nout -> do
        c_tc_bwt_decode_sym ctx syms (fromIntegral n) out nout >>= check ctx
        k <- fromIntegral <$> peek nout
        BS.packCStringLen (castPtr out, k)

-- | MTF.hs:117-122
bytestringBWTToMTFB :: BWT Word8 -> MTF BS.ByteString
bytestringBWTToMTFB (BWT s)
  | DS.null s = MTF (DS.Empty, DS.Empty)
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map symOf (toList s)) $ \n syms ->
      allocaArray n $ \idx -> allocaArray 257 $ \fl -> alloca $ \sg -> do
        c_tc_mtf_encode_sym ctx syms (fromIntegral n) idx fl sg >>= check ctx
        sigma <- fromIntegral <$> peek sg
        is <- peekArray n idx
        ls <- peekArray sigma fl
        pure $ MTF (DS.fromList (map fromIntegral is), DS.fromList (map (fmap BS.singleton . ofSym) ls))

-- | MTF.hs:240-245
bytestringBWTFromMTFB :: MTF BS.ByteString -> BWT BS.ByteString
bytestringBWTFromMTFB (MTF (is, ls))
  | DS.null is || DS.null ls = BWT DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map fromIntegral (toList is)) $ \n idx ->
      withArrayLen (map (symOf . fmap BS.head) (toList ls)) $ \k fl ->
      allocaArray n $ \out -> do
        c_tc_mtf_decode ctx idx (fromIntegral n) fl (fromIntegral k) out >>= check ctx
        BWT . DS.fromList . map (fmap BS.singleton . ofSym) <$> peekArray n out

-- | RLE.hs:117-123; counts rendered with `show` exactly as RLE/Internal.hs:128 does
bytestringBWTToRLEB :: BWT Word8 -> RLE BS.ByteString
bytestringBWTToRLEB (BWT s)
  | DS.null s = RLE DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      withArrayLen (map symOf (toList s)) $ \n syms ->
      let cap = 2 * n + 2 in
      allocaArray cap $ \cs -> allocaArray cap $ \vs -> with (fromIntegral cap) $ \nr -> do
        c_tc_rle_encode_sym ctx syms (fromIntegral n) cs vs nr >>= check ctx
        k <- fromIntegral <$> peek nr
        counts <- peekArray k cs
        vals <- peekArray k vs
        pure . RLE . DS.fromList . concat $
          [ [Just (BSC8.pack (show c)), fmap BS.singleton (ofSym v)] | (c, v) <- zip counts vals ]

-- | RLE.hs:237-241 (pairs; `(Just _, Nothing)` is one Nothing; an odd tail is ignored)
bytestringBWTFromRLEB :: RLE BS.ByteString -> BWT BS.ByteString
bytestringBWTFromRLEB (RLE s)
  | DS.null s = BWT DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx -> do
      let prs = pairs (toList s)
          cnt (Just c, Just _) = max 0 (read (BSC8.unpack c)) :: Int   -- `read` throws as the reference does
          cnt _ = 1
          sym (_, Nothing) = -1
          sym (_, Just b) = fromIntegral (BS.head b)
          total = sum [ if sym p < 0 then 1 else cnt p | p <- prs ]
      withArrayLen (map (fromIntegral . cnt) prs) $ \k cs ->
        withArray (map sym prs) $ \vs ->
        allocaArray (total + 1) $ \out -> with (fromIntegral (total + 1)) $ \nn -> do
          c_tc_rle_decode ctx cs vs (fromIntegral k) out nn >>= check ctx
          m <- fromIntegral <$> peek nn
          BWT . DS.fromList . map (fmap BS.singleton . ofSym) <$> peekArray m out
  where pairs (a : b : r) = (a, b) : pairs r
        pairs _ = []

-- | FMIndex.hs:362-379: one batched device call; 0 stands for Nothing; order preserved
bytestringFMIndexCountS :: [BS.ByteString] -> BS.ByteString -> Seq (BS.ByteString, Maybe Int)
bytestringFMIndexCountS pats input
  | null pats || BS.null input = DS.Empty
  | otherwise = unsafePerformIO $ withCtx $ \ctx ->
      BSU.unsafeUseAsCStringLen input $ \(p, n) -> alloca $ \ph -> do
        c_tc_fm_build ctx (castPtr p) (fromIntegral n) ph >>= check ctx
        fm <- peek ph
        let flat = BS.concat pats `BS.snoc` 0
            offs = scanl (+) 0 (map (fromIntegral . BS.length) pats)
        r <- BSU.unsafeUseAsCString flat $ \fp -> withArray offs $ \op ->
               allocaArray (length pats) $ \out -> do
                 rc <- c_tc_fm_count ctx fm (castPtr fp) op (fromIntegral (length pats)) out
                 c_tc_fm_free fm
                 check ctx rc
                 peekArray (length pats) out
        pure . DS.fromList $ [ (q, if c == 0 then Nothing else Just (fromIntegral c)) | (q, c) <- zip pats r ]

-- | FMIndex.hs:411-432: the parListChunk spark pool is the one batched launch above
bytestringFMIndexCountP :: [BS.ByteString] -> BS.ByteString -> IO (Seq (BS.ByteString, Maybe Int))
bytestringFMIndexCountP pats input = pure (bytestringFMIndexCountS pats input)
